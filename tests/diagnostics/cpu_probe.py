import time, torch, numpy as np, sys
sys.path.insert(0, '.')
from sleap_amd.nn.architectures import build_unet_model_config, he_normal_weights
from oracle.keras_graph import KerasGraph
cfg, shapes = build_unet_model_config((1024,1024,1),16,2,32,4,True,True,heads=[("MultiInstanceConfmapsHead",13,4),("PartAffinityFieldsHead",24,8)])
g = KerasGraph(cfg, he_normal_weights(shapes))
x = np.random.rand(1,1024,1024,1).astype(np.float32)
for nt in (16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    g(x)
    t=time.time(); g(x); print(nt, 'threads', time.time()-t, 's/frame', flush=True)
