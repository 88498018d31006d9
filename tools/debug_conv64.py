"""Debug aid (round 5): is kfn_conv3x3_c64_f16 (conv64_rows_kernel) reproducible from launch to launch while another stream
keeps the memory system busy?  MB_LIB=<variant .so> selects a debugging build (tools/mb/build_c64.sh: KFN_STORE_PAD = -1 is
round 4's kernel, without the wait states behind its 16-byte buffer stores -- kfn_common.h buffer_store_b128).
  python tools/debug_conv64.py [REPS] [quick]      quick: only 5x540x960 with the side-stream load"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from kfnet_amd import _lib
if os.environ.get('MB_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MB_LIB'])
from kfnet_amd.graph import pack_conv64_rows_kernel
lib = _lib.load()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
QUICK = len(sys.argv) > 2 and sys.argv[2] == 'quick'
rng = np.random.default_rng(3)
for (n, h, w) in ([(5, 540, 960)] if QUICK else [(2, 540, 960), (5, 540, 960), (1, 68, 120)]):
    x = torch.from_numpy(np.maximum(rng.normal(size=(n, h, w, 64)), 0).astype(np.float16)).cuda()
    wt = (rng.normal(size=(3, 3, 64, 64)) * np.sqrt(2.0 / 576)).astype(np.float32)
    wp = torch.from_numpy(pack_conv64_rows_kernel(wt)).cuda()
    b = torch.from_numpy(rng.normal(size=64).astype(np.float32)).cuda()
    d = _lib.ConvDesc(N=n, H=h, W=w, Cin=64, ldx=64, Cout=64, cout_pad=64, ldy=64, kh=3, kw=3, stride=1, relu=1, operand_dtype=1,
                      x_dtype=1, y_dtype=1)
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    big_a = torch.randn(64 << 20, device='cuda')
    big_b = torch.empty_like(big_a)
    for load in ((True,) if QUICK else (False, True)):
        ref = ref_dev = None
        bad = 0
        first_bad = None
        for r in range(REPS):
            y = torch.full((n * h * w, 64), -3.0, dtype=torch.float16, device='cuda')
            torch.cuda.synchronize()
            if load:
                with torch.cuda.stream(side):
                    for _ in range(6):
                        big_b.copy_(big_a)          # 512 MB of HBM traffic beside the convolution
            _lib.check(lib.kfn_conv3x3_c64_f16(C.byref(d), x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), main.cuda_stream), 'c64')
            torch.cuda.synchronize()
            if ref is None:
                ref_dev = y
                ref = y.cpu().numpy()
            elif not torch.equal(y, ref_dev):
                out = y.cpu().numpy()
                bad += 1
                if first_bad is None or bad <= 3:
                    idx = np.argwhere(out != ref)
                    rows = sorted(set(int(i[0]) // w % h for i in idx))
                    first_bad = 'rep %d: %d elements, image rows %s...' % (r, len(idx), rows[:10])
                    o4, r4 = out.reshape(n, h, w, 64), ref.reshape(n, h, w, 64)
                    pos = np.argwhere(o4 != r4)
                    print('   rep %d: %d differing elements' % (r, len(pos)))
                    seen = set()
                    for (a, yy, xx, cc) in pos:
                        key = (a, yy, xx, cc // 8)
                        if key in seen:
                            continue
                        seen.add(key)
                        c0 = 8 * (cc // 8)
                        got8, ref8 = o4[a, yy, xx, c0:c0 + 8], r4[a, yy, xx, c0:c0 + 8]
                        # does the wrong piece equal the REFERENCE's piece somewhere near (another row / pixel / image)?
                        match = None
                        for dy in range(-4, 5):
                            for dx in (-2, -1, 0, 1, 2):
                                y2, x2 = yy + dy, xx + dx
                                if (dy or dx) and 0 <= y2 < h and 0 <= x2 < w and np.array_equal(r4[a, y2, x2, c0:c0 + 8], got8):
                                    match = (dy, dx)
                        print('      img %d row %d px %d ch %d-%d: got %s want %s%s' % (a, yy, xx, c0, c0 + 7, np.round(got8.astype(np.float32), 3).tolist(),
                              np.round(ref8.astype(np.float32), 3).tolist(), '  == reference piece at (dy,dx)=%s' % (match,) if match else ''))
                        if len(seen) >= 12:
                            break
        print('%dx%dx%d  side-stream load %-5s: %d of %d launches differ from the first  %s' % (n, h, w, load, bad, REPS - 1, first_bad or ''))
