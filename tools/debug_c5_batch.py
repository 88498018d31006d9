"""Debug aid (round 5): test_config5_batch_independence_at_full_size failed once with a 4e-5 difference in the LAST frame of
the second sequence.  Which buffer differs between the batch-5 engine (both sequences in one scan launch) and the batch-2
engine (one sequence per launch) -- the scan inputs (heavy phase) or only the records (scan)?  And is it stable from run to run?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kfnet_amd.engine import KFNetEngine
from kfnet_amd.synth import synthetic_sequence
from kfnet_amd.weights import synthetic_weights

W = synthetic_weights(1234)
seqs = np.stack([synthetic_sequence(5, 540, 960, seed=11 + s) for s in range(2)])
dev = torch.from_numpy(seqs).cuda()
T4 = np.eye(4, dtype=np.float32)


def run(batch, sel, reps=3):
    eng = KFNetEngine(W, image_size=(540, 960), batch=batch, transform=T4, reset_period=500, max_chunk=10, conv_operands='f16')
    outs = []
    for r in range(reps):
        rec = eng.process_sequences(dev[sel]).cpu().numpy().copy()
        n = rec.shape[0] * rec.shape[1]
        d = {k: v.copy() for k, v in eng.debug(n).items()}
        outs.append((rec, d))
    del eng
    torch.cuda.empty_cache()
    return outs


def cmp(name, a, b):
    if np.array_equal(a, b):
        return True
    idx = np.argwhere(a != b)
    print('   %s differs: %d elements, first at %s (a=%r b=%r), frames touched %s' % (name, len(idx), tuple(idx[0]), a[tuple(idx[0])], b[tuple(idx[0])], sorted(set(int(i[0]) for i in idx))[:8]))
    return False


both = run(5, slice(0, 2))
for r in range(1, len(both)):
    print('batch-5 engine, repetition %d vs 0: records %s' % (r, 'same' if cmp('rec', both[r][0], both[0][0]) else 'DIFFER'))
for s in range(2):
    alone = run(2, slice(s, s + 1))
    for r in range(1, len(alone)):
        print('batch-2 engine seq %d, repetition %d vs 0: records %s' % (s, r, 'same' if cmp('rec', alone[r][0], alone[0][0]) else 'DIFFER'))
    ok = cmp('records', alone[0][0][0], both[0][0][s])
    print('sequence %d: records alone vs both: %s' % (s, 'same' if ok else 'DIFFER'))
    for k in ('flow', 'sigma_trans', 'meas'):
        a = alone[0][1][k][:5]
        b = both[0][1][k][5 * s:5 * s + 5]
        print('   scan input %-12s %s' % (k, 'same' if cmp(k, a, b) else 'DIFFER'))
