"""-m "not gpu": the library's native PNG decoder (`kfn_decode_png_rgb8`, csrc/kfn_png.hip -- host code, runs without a GPU)
against PIL, which is what `kfnet_amd.pipeline.decode_image` restates of tf.image.decode_png(channels=3)
(KFNet/train.py:213-217): every colour type and bit depth PIL writes, every scanline filter of the PNG specification on
hand-built files, the files it hands back to PIL (16-bit), the errors it must name, and the ChunkLoader on both decoders."""
import os
import struct
import zlib

import numpy as np
import pytest

from kfnet_amd.pipeline import ChunkLoader, decode_image, decode_png_batch

H, W = 13, 21     # odd sizes: packed rows of 1 / 2 / 4-bit samples end inside a byte


def _native(paths, size=(H, W), threads=3):
    dst = np.full((len(paths), size[0], size[1], 3), 77, np.uint8)
    decode_png_batch([str(p) for p in paths], dst, size, threads)
    return dst


def _pil(paths, size=(H, W)):
    return np.stack([decode_image(str(p), size) for p in paths])


@pytest.mark.parametrize('mode', ['RGB', 'RGBA', 'L', 'LA', 'P', '1'])
def test_modes_written_by_pil(tmp_path, mode):
    from PIL import Image
    rng = np.random.default_rng(len(mode))
    paths = []
    for k, level in enumerate((0, 1, 6, 9)):
        if mode == '1':
            im = Image.fromarray((rng.integers(0, 2, size=(H, W)) * 255).astype(np.uint8)).convert('1')
        elif mode == 'P':
            im = Image.fromarray(rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)).quantize(colors=(2, 5, 16, 200)[k])
        else:
            ch = {'RGB': 3, 'RGBA': 4, 'L': 1, 'LA': 2}[mode]
            a = rng.integers(0, 256, size=(H, W, ch), dtype=np.uint8)
            im = Image.fromarray(a[..., 0] if ch == 1 else a, mode)
        p = tmp_path / ('%s_%d.png' % (mode, k))
        im.save(p, compress_level=level)
        paths.append(p)
    assert np.array_equal(_native(paths), _pil(paths))


def _chunk(tag, body):
    return struct.pack('>I', len(body)) + tag + body + struct.pack('>I', zlib.crc32(tag + body) & 0xffffffff)


def _filter_row(ft, cur, prev, bpp):
    """forward filter of the PNG specification (section 9.2) on one row of bytes"""
    out = bytearray(len(cur))
    for i in range(len(cur)):
        a = cur[i - bpp] if i >= bpp else 0
        b = prev[i] if prev is not None else 0
        c = prev[i - bpp] if (prev is not None and i >= bpp) else 0
        if ft == 0:
            pred = 0
        elif ft == 1:
            pred = a
        elif ft == 2:
            pred = b
        elif ft == 3:
            pred = (a + b) // 2
        else:
            p = a + b - c
            pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
            pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
        out[i] = (cur[i] - pred) & 255
    return bytes(out)


def _write_png(path, rows, ctype, depth, filters, palette=None, idat_pieces=1, width=W):
    """rows: list of packed scanlines (bytes); filters: one filter type per row"""
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, channels * depth // 8)
    raw, prev = b'', None
    for r, ft in zip(rows, filters):
        raw += bytes([ft]) + _filter_row(ft, r, prev, bpp)
        prev = r
    z = zlib.compress(raw, 6)
    cut = [len(z) * k // idat_pieces for k in range(idat_pieces + 1)]
    data = b'\x89PNG\r\n\x1a\n' + _chunk(b'IHDR', struct.pack('>IIBBBBB', width, len(rows), depth, ctype, 0, 0, 0))
    data += _chunk(b'tEXt', b'Comment\x00hand-built')          # an ancillary chunk in front of the data
    if palette is not None:
        data += _chunk(b'PLTE', bytes(palette))
    for k in range(idat_pieces):
        data += _chunk(b'IDAT', z[cut[k]:cut[k + 1]])
    data += _chunk(b'IEND', b'')
    with open(path, 'wb') as f:
        f.write(data)


def _pack(samples, depth):
    """[W] integer samples -> packed bytes, most significant bits first"""
    if depth == 8:
        return bytes(int(v) for v in samples)
    per = 8 // depth
    out = bytearray((len(samples) + per - 1) // per)
    for x, v in enumerate(samples):
        out[x // per] |= int(v) << ((per - 1 - x % per) * depth)
    return bytes(out)


@pytest.mark.parametrize('ctype,depth', [(2, 8), (6, 8), (0, 8), (4, 8), (0, 4), (0, 2), (0, 1), (3, 8), (3, 4), (3, 2), (3, 1)])
def test_every_filter_type_on_hand_built_files(tmp_path, ctype, depth):
    """Rows cycle through filters 0..4 (and start with each of them once: the first row has no row above), the image data
    is split over several IDAT chunks; PIL reads the same files."""
    rng = np.random.default_rng(ctype * 10 + depth)
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    palette = None
    if ctype == 3:
        palette = rng.integers(0, 256, size=3 * (1 << depth), dtype=np.uint8).tolist()
    paths = []
    for first in range(5):
        smooth = np.cumsum(rng.integers(-3, 4, size=(H, W * channels)), axis=1) % (1 << depth)      # filters matter on smooth data
        rows = [_pack(smooth[y], depth) for y in range(H)]
        p = tmp_path / ('f_%d_%d_%d.png' % (ctype, depth, first))
        _write_png(p, rows, ctype, depth, [(first + y) % 5 for y in range(H)], palette, idat_pieces=1 + first)
        paths.append(p)
    assert np.array_equal(_native(paths), _pil(paths))


def test_sixteen_bit_files_go_back_to_pil_and_errors_name_the_file(tmp_path):
    from PIL import Image
    from kfnet_amd import _lib
    import ctypes as C
    rng = np.random.default_rng(0)
    ok = tmp_path / 'ok.png'
    Image.fromarray(rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)).save(ok)
    deep = tmp_path / 'deep.png'
    Image.fromarray(rng.integers(0, 65536, size=(H, W)).astype(np.uint16)).save(deep)       # 16-bit gray
    lib = _lib.load()
    paths = [str(ok), str(deep)]
    arr = (C.c_char_p * 2)(*[os.fsencode(p) for p in paths])
    st = (C.c_int * 2)()
    dst = np.full((2, H, W, 3), 9, np.uint8)
    assert lib.kfn_decode_png_rgb8(arr, 2, H, W, dst.ctypes.data, st, 2) == 0
    assert list(st) == [_lib.PNG_OK, _lib.PNG_UNSUPPORTED] and np.all(dst[1] == 9)          # untouched: the host's fallback takes it
    assert np.array_equal(_native(paths), _pil(paths))                                      # ... and decode_png_batch does
    # errors: wrong size, truncated data, not a PNG, missing -- each names its file; the good file of the call is still decoded
    small = tmp_path / 'small.png'
    Image.fromarray(rng.integers(0, 256, size=(H - 1, W, 3), dtype=np.uint8)).save(small)
    cut = tmp_path / 'cut.png'
    cut.write_bytes(ok.read_bytes()[:-40])
    junk = tmp_path / 'junk.png'
    junk.write_bytes(b'not a png at all' * 8)           # (no PNG signature: handed to PIL, whose error is re-raised as ValueError)
    for bad in (small, cut, junk, tmp_path / 'missing.png'):
        with pytest.raises(ValueError) as e:
            _native([ok, bad])
        assert os.path.basename(str(bad)) in str(e.value)
    with pytest.raises(ValueError) as e:
        _native([small])
    assert '%dx%d' % (H - 1, W) in str(e.value) and 'expected %dx%d' % (H, W) in str(e.value)       # decode_image's message
    # a file that is not a PNG but that PIL reads (the reference's lists hold PNGs; nothing forbids a JPEG) is decoded, not refused
    jpg = tmp_path / 'photo.jpg'
    Image.fromarray(rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)).save(jpg, quality=95)
    assert np.array_equal(_native([ok, jpg]), _pil([ok, jpg]))
    st3 = (C.c_int * 3)()
    arr3 = (C.c_char_p * 3)(*[os.fsencode(str(p)) for p in (ok, cut, ok)])
    dst3 = np.zeros((3, H, W, 3), np.uint8)
    assert lib.kfn_decode_png_rgb8(arr3, 3, H, W, dst3.ctypes.data, st3, 0) == -1
    assert list(st3) == [_lib.PNG_OK, _lib.PNG_ERROR, _lib.PNG_OK] and np.array_equal(dst3[0], dst3[2])
    assert lib.kfn_decode_png_rgb8(arr3, 0, H, W, None, None, 4) == 0                               # empty chunk
    assert lib.kfn_decode_png_rgb8(arr3, 3, 0, W, dst3.ctypes.data, None, 4) == -1


def test_chunk_loader_is_the_same_on_both_decoders(tmp_path):
    from PIL import Image
    from kfnet_amd.synth import synthetic_sequence
    frames = synthetic_sequence(11, 48, 64, seed=3)
    paths = []
    for i in range(frames.shape[0]):
        p = str(tmp_path / ('frame-%03d.png' % i))
        Image.fromarray(frames[i]).save(p, compress_level=(1, 6)[i % 2])
        paths.append(p)
    got = {}
    for native in (True, False):
        loader = ChunkLoader(paths, (48, 64), chunk=4, workers=3, pinned=False, first_chunk=(1, 2), native=native)
        assert loader.native == native
        got[native] = [(lo, host.numpy().copy()) for lo, host in loader]
    assert [g[0] for g in got[True]] == [g[0] for g in got[False]] == [0, 1, 3, 7]
    for a, b in zip(got[True], got[False]):
        assert np.array_equal(a[1], b[1])
    assert np.array_equal(np.concatenate([g[1] for g in got[True]]), frames)
    # a custom decode callable keeps the Python pool
    calls = []
    def dec(path, size):
        calls.append(path)
        return decode_image(path, size)
    out = [h.numpy().copy() for _, h in ChunkLoader(paths, (48, 64), chunk=4, workers=2, pinned=False, decode=dec)]
    assert sorted(calls) == sorted(paths) and np.array_equal(np.concatenate(out), frames)


def test_mutated_files_never_crash_the_library(tmp_path):
    """600 byte-level mutations of valid files (overwritten / deleted / inserted bytes, header and chunk-length fields among
    them) through ONE call on several threads: every file comes back OK, UNSUPPORTED or ERROR -- native code that parses
    files must not fall over them -- and an OK frame of an untouched seed equals PIL's."""
    from PIL import Image
    from kfnet_amd import _lib
    import ctypes as C
    lib = _lib.load()
    rng = np.random.default_rng(0)
    seeds = []
    for mode, ch in (('RGB', 3), ('RGBA', 4), ('L', 1), ('LA', 2)):
        a = rng.integers(0, 256, size=(H, W, ch), dtype=np.uint8)
        p = tmp_path / ('seed_%s.png' % mode)
        Image.fromarray(a[..., 0] if ch == 1 else a, mode).save(p)
        seeds.append(p.read_bytes())
    p = tmp_path / 'seed_P.png'
    Image.fromarray(rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)).quantize(colors=7).save(p)
    seeds.append(p.read_bytes())
    paths = [str(tmp_path / 'seed_RGB.png')]
    for i in range(600):
        b = bytearray(seeds[i % len(seeds)])
        for _ in range(int(rng.integers(1, 6))):
            op, pos = int(rng.integers(0, 4)), int(rng.integers(8, len(b)))
            if op == 0:
                b[pos] = int(rng.integers(0, 256))
            elif op == 1:
                del b[pos:pos + int(rng.integers(1, 20))]
            elif op == 2:
                b[pos:pos] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 20)), dtype=np.uint8))
            else:
                b[int(rng.integers(8, min(len(b), 40)))] = int(rng.integers(0, 256))       # IHDR fields / first chunk lengths
        q = tmp_path / ('m%04d.png' % i)
        q.write_bytes(bytes(b))
        paths.append(str(q))
    n = len(paths)
    arr = (C.c_char_p * n)(*[os.fsencode(q) for q in paths])
    st = (C.c_int * n)()
    dst = np.zeros((n, H, W, 3), np.uint8)
    rc = lib.kfn_decode_png_rgb8(arr, n, H, W, dst.ctypes.data, st, 4)
    assert rc in (0, -1) and set(st) <= {_lib.PNG_OK, _lib.PNG_UNSUPPORTED, _lib.PNG_ERROR}
    assert st[0] == _lib.PNG_OK and np.array_equal(dst[0], decode_image(paths[0], (H, W)))
    assert list(st).count(_lib.PNG_ERROR) > 400          # most mutations are caught (bad inflate / sizes / chunks)


def test_crc_and_chunk_order_are_checked(tmp_path):
    """a damaged IHDR / PLTE / IDAT byte (CRC left as it was), a first chunk that is not IHDR and a second IHDR are
    errors, as for libpng (the reference's tf.image.decode_png) and PIL; a damaged ANCILLARY chunk is not."""
    from kfnet_amd import _lib
    import ctypes as C
    lib = _lib.load()
    rng = np.random.default_rng(3)
    rows = [_pack(rng.integers(0, 4, size=W), 2) for _ in range(H)]
    pal = rng.integers(0, 256, size=12, dtype=np.uint8).tolist()
    good = tmp_path / 'good.png'
    _write_png(good, rows, 3, 2, [y % 5 for y in range(H)], pal, idat_pieces=2)
    data = good.read_bytes()

    def chunks(b):
        out, pos = [], 8
        while pos < len(b):
            n = struct.unpack('>I', b[pos:pos + 4])[0]
            out.append((b[pos + 4:pos + 8], pos, n))
            pos += 12 + n
        return out
    at = {}
    for t, pos, n in chunks(data):
        at.setdefault(t, (pos, n))
    cases = {}
    for t in (b'IHDR', b'PLTE', b'IDAT', b'tEXt'):
        pos, n = at[t]
        b = bytearray(data)
        b[pos + 8 + n - 1] ^= 0x01             # last body byte of the chunk, CRC untouched
        cases[t.decode()] = bytes(b)
    ih_pos, ih_n = at[b'IHDR']
    ihdr = data[ih_pos:ih_pos + 12 + ih_n]
    tx_pos, tx_n = at[b'tEXt']
    text = data[tx_pos:tx_pos + 12 + tx_n]
    cases['text_first'] = data[:8] + text + ihdr + data[tx_pos + 12 + tx_n:]
    cases['two_ihdr'] = data[:8] + ihdr + ihdr + data[8 + len(ihdr):]
    names = ['good'] + sorted(cases)
    paths = [str(good)]
    for k in names[1:]:
        q = tmp_path / (k + '.png')
        q.write_bytes(cases[k])
        paths.append(str(q))
    n = len(paths)
    arr = (C.c_char_p * n)(*[os.fsencode(q) for q in paths])
    st = (C.c_int * n)()
    dst = np.zeros((n, H, W, 3), np.uint8)
    lib.kfn_decode_png_rgb8(arr, n, H, W, dst.ctypes.data, st, 2)
    got = dict(zip(names, list(st)))
    assert got['good'] == _lib.PNG_OK and got['tEXt'] == _lib.PNG_OK
    assert np.array_equal(dst[names.index('tEXt')], dst[0])
    for k in ('IHDR', 'PLTE', 'IDAT', 'text_first', 'two_ihdr'):
        assert got[k] == _lib.PNG_ERROR, (k, got[k])
    assert b'CRC' in lib.kfn_last_error() or b'IHDR' in lib.kfn_last_error()
