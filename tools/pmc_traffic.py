#!/usr/bin/env python
"""Derive per-kernel HBM traffic per launch from rocprofv3 PMC passes (rocpd sqlite DBs).

    python tools/pmc_traffic.py <fetch_db> <write_db> [out.json] [--only KERNEL --suffix TEXT --into EXISTING.json]

(--only/--suffix/--into: keep one kernel of a dedicated run, e.g. the S=256 x T=64 Kalman-scan
roofline launch of tools/kalman_roofline.py, and merge it as "KERNEL<suffix>" into the round's file.)

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  Per
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts 128-byte
requests at 64 B for wide coalesced streams, so the read side is DOUBLED here; WRITE_SIZE
is taken as is (uncalibrated).  Output: {kernel_tag: {fetch_bytes, write_bytes, hbm_bytes,
launches}} averaged per launch, keyed like bench.py's roofline.kernel."""
import json
import re
import sqlite3
import sys


def tag(name):
    m = re.search(r'(conv_mfma_kernel<[^>]*>)', name)
    if m:
        return m.group(1)
    m = re.search(r'([a-z_0-9]+_kernel)', name)
    return m.group(1) if m else name


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, n, avg in c.execute("select kernel_name, count(*), avg(value) from counters_collection "
                                  "where counter_name=? group by kernel_name", (counter,)):
        out[tag(name)] = (n, avg)
    return out


def main():
    argv = list(sys.argv)
    opts = {}
    for flag in ('--only', '--suffix', '--into'):
        if flag in argv:
            i = argv.index(flag)
            opts[flag] = argv[i + 1]
            del argv[i:i + 2]
    sys.argv = argv
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    res = {}
    if '--into' in opts:
        try:
            res = json.load(open(opts['--into']))
        except (OSError, ValueError):
            res = {}
    for k in sorted(set(fetch) | set(write)):
        if '--only' in opts and k != opts['--only']:
            continue
        f = fetch.get(k, (0, 0.0))
        w = write.get(k, (0, 0.0))
        fb, wb = 2.0 * f[1] * 1024.0, w[1] * 1024.0
        res[k + opts.get('--suffix', '')] = {'launches_sampled': int(max(f[0], w[0])), 'fetch_bytes_per_launch': int(fb),
                  'write_bytes_per_launch': int(wb), 'hbm_bytes_per_launch': int(fb + wb),
                  'note': 'FETCH_SIZE x2 (gfx950 128B-request correction), WRITE_SIZE uncalibrated'}
    res.setdefault('__sampled__', {'command': 'bench.py --steps 64 --batch 32 (tools/profile_round.sh)', 'tower_batch': 32})
    s = json.dumps(res, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], 'w').write(s + '\n')
    print(s)


if __name__ == '__main__':
    main()
