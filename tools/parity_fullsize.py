"""One-off full-size parity run: T frames of the BASELINE 480x640 sequence through the HIP path
and through the torch-CPU fp32 restatement (towers once per frame), with a reset in the middle.
usage: python tools/parity_fullsize.py [T] [out.json]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kfnet_amd.engine import KFNetEngine
from kfnet_amd.synth import synthetic_sequence, synthetic_transform
from kfnet_amd.weights import synthetic_weights
from oracle import kfnet_oracle as O
from oracle import kfnet_oracle_torch as OT
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = synthetic_weights(1234)
T4 = O.get_transform(synthetic_transform())
imgs = synthetic_sequence(T, 480, 640, seed=1)
period = 40                      # one reset inside the run
t0 = time.time()
ref = OT.eval_sequence(imgs, W, T4, reset_period=period)
t_cpu = time.time() - t0
out = {'frames': T, 'reset_period': period, 'cpu_seconds': round(t_cpu, 1)}
for mode in ('f32', 'f16x3'):
    eng = KFNetEngine(W, image_size=(480, 640), batch=17, transform=T4, reset_period=period, max_chunk=T,
                      conv_operands=mode)
    rec = eng.process(eng.upload_frames(imgs)).cpu().numpy()
    dc = np.abs(rec[..., :3] - ref[..., :3]).reshape(T, -1).max(1)
    dr = (np.abs(rec[..., 3] - ref[..., 3]) / np.abs(ref[..., 3])).reshape(T, -1).max(1)
    out[mode] = {'coord_max_abs': float(dc.max()), 'conf_max_rel': float(dr.max()),
                 'coord_max_abs_first_last_8': [float(v) for v in list(dc[:8]) + list(dc[-8:])],
                 'worst_frame': int(dc.argmax())}
    del eng
out['tolerance'] = 'coord max-abs <= 1e-4, confidence max-rel <= 1e-4'
s = json.dumps(out, indent=1)
print(s)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(s + '\n')
