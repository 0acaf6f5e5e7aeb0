"""Data formats either side of the inference path."""
