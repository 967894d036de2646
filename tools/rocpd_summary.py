#!/usr/bin/env python
"""Turn a rocprofv3 rocpd database (--kernel-trace --stats) into the small
per-kernel summary committed under profiles/ (name, calls, total, avg, min,
max in microseconds, share)."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        'select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), '
        'max(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s '
        'on d.kernel_id = s.id group by s.kernel_name order by 3 desc').fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ['kernel,calls,total_us,avg_us,min_us,max_us,percent']
    for name, calls, tot, mn, mx in rows:
        lines.append(f'"{name}",{calls},{tot/1e3:.1f},{tot/calls/1e3:.2f},{mn/1e3:.2f},'
                     f'{mx/1e3:.2f},{100*tot/total:.2f}')
    text = '\n'.join(lines) + '\n'
    if out_path:
        open(out_path, 'w').write(text)
    print(text)


if __name__ == '__main__':
    main(*sys.argv[1:3])
