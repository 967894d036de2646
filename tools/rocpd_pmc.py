#!/usr/bin/env python
"""Per-kernel average of each PMC counter from a rocprofv3 rocpd database."""
import sqlite3
import sys
from collections import defaultdict


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    cols = [c[1] for c in db.execute("pragma table_info('rocpd_pmc_event')")]
    pcols = [c[1] for c in db.execute("pragma table_info('rocpd_info_pmc')")]
    ecols = [c[1] for c in db.execute("pragma table_info('rocpd_event')")]
    # pmc_event.event_id -> rocpd_event.id ; kernel_dispatch.event_id -> same
    q = ('select s.kernel_name, i.name, sum(p.value), count(distinct d.id) '
         'from rocpd_pmc_event p join rocpd_info_pmc i on p.pmc_id = i.id '
         'join rocpd_kernel_dispatch d on d.event_id = p.event_id '
         'join rocpd_info_kernel_symbol s on d.kernel_id = s.id '
         'group by s.kernel_name, i.name')
    try:
        rows = db.execute(q).fetchall()
    except Exception as e:
        print('schema:', cols, pcols, ecols, file=sys.stderr)
        raise
    tab = defaultdict(dict)
    calls = {}
    for k, c, v, n in rows:
        tab[k][c] = v / max(n, 1)
        calls[k] = n
    counters = sorted({c for k in tab for c in tab[k]})
    lines = ['kernel,calls,' + ','.join(counters)]
    for k in sorted(tab, key=lambda k: -tab[k].get('SQ_BUSY_CYCLES', 0)):
        lines.append(f'"{k}",{calls[k]},' + ','.join(f'{tab[k].get(c, 0):.0f}' for c in counters))
    text = '\n'.join(lines) + '\n'
    if out_path:
        open(out_path, 'w').write(text)
    print(text)


if __name__ == '__main__':
    main(*sys.argv[1:3])
