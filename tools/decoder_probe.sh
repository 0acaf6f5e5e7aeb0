# Is there ANY H.264 decoder on the GPU box? Prints what it finds; the output is kept under profiles/.
echo "== executables"; for x in ffmpeg ffprobe gst-launch-1.0 mplayer mpv vlc avconv x264; do printf "%s: " $x; which $x 2>/dev/null || echo none; done
echo "== python modules"; python - <<'PY'
for m in ['cv2', 'av', 'imageio', 'imageio_ffmpeg', 'torchvision', 'torchvision.io', 'decord', 'skvideo', 'moviepy', 'torchcodec', 'torchaudio', 'nvidia.dali', 'rocdecode', 'pyrocdecode', 'gi']:
    try:
        __import__(m)
        print(m, 'importable')
    except Exception as e:
        print(m, 'absent:', type(e).__name__)
try:
    import gi
    gi.require_version('Gst', '1.0')
    from gi.repository import Gst  # noqa: F401
    print('GStreamer typelib present')
except Exception as e:
    print('GStreamer typelib absent:', e)
PY
echo "== shared libraries"; (ldconfig -p 2>/dev/null; ls /opt/rocm/lib /usr/lib/x86_64-linux-gnu /usr/local/lib 2>/dev/null) | grep -i -E 'avcodec|avformat|swscale|libva\.|libva-|gstlibav|openh264|x264|rocdecode|rocjpeg|vdpau|libde265|mfx|vpl' | sort -u | head -20; echo "(end of list)"
echo "== device nodes"; ls /dev/dri 2>/dev/null; ls /dev/video* 2>/dev/null || echo "no /dev/video*"
