"""Host-side mirror of the reference's `sleap.nn` surface for the bottom-up inference path."""
