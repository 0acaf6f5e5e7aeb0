"""Two encodings of the SAME video decoded side by side (the reference holds centered_pair_low_quality.mp4 at QP ~21 and
centered_pair_small.mp4 at QP ~9-20: different motion vectors, residuals and weights): PSNR between the two decodes per display
frame. Both agree on key frames by construction (the intra module, pinned to TensorFlow's golden); if the inter pictures were
decoded wrongly (interpolation, weights, edge filter -- arithmetic the parser's self-checks cannot see) they would disagree far more
than the key frames do. Result: profiles/r06_h264_cross_check.txt.      python tools/h264_cross_check.py [n_frames]"""
import sys, numpy as np
sys.path.insert(0, '.')
from sleap_amd.io import _h264 as D
a = D.H264Reader('tests/golden/video/centered_pair_low_quality.mp4')
b = D.H264Reader('tests/golden/video/centered_pair_small.mp4')
def psnr(x,y):
    mse=np.mean((x.astype(float)-y.astype(float))**2); return 10*np.log10(255**2/mse)
N = int(sys.argv[1]) if len(sys.argv) > 1 else len(a)
for k in range(N):
    ya=a.frame(k)[0]; yb=b.frame(k)[0]
    sa=a.track.display_order[k]
    print(k, 'sample', sa, 'cross PSNR %.2f' % psnr(ya,yb), flush=True)
