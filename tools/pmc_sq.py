#!/usr/bin/env python
"""Per-kernel averages of SQ/GRBM counters from rocprofv3 PMC passes (rocpd sqlite DBs) and
the derived MFMA utilisation.

    python tools/pmc_sq.py out.json pass1.db [pass2.db ...]

MfmaUtil_pct = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): the share
of SIMD-cycles in which the MFMA pipe was busy while the kernel ran."""
import json
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from pmc_traffic import tag  # noqa: E402


def main():
    out = {}
    for db in sys.argv[2:]:
        c = sqlite3.connect(db)
        for name, counter, avg in c.execute("select kernel_name, counter_name, avg(value) from counters_collection "
                                            "group by kernel_name, counter_name"):
            out.setdefault(tag(name), {})[counter] = avg
    for k, v in out.items():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and v.get('GRBM_GUI_ACTIVE'):
            v['MfmaUtil_pct'] = round(100.0 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0), 2)
    json.dump(out, open(sys.argv[1], 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
