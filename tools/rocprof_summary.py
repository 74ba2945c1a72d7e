"""Summarise rocprofv3 sqlite outputs (gpurun_out/...) into small text/JSON files under profiles/.

    python tools/rocprof_summary.py stats  gpurun_out/prof_r1/bench_results.db  profiles/r1_bench_kernel_stats.txt
    python tools/rocprof_summary.py pmc    gpurun_out/pmc_fetch/bench_results.db gpurun_out/pmc_write/bench_results.db profiles/r1_bench_pmc_hbm.json
"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    return name if len(name) < 90 else name[:87] + '...'


def stats(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    with open(out, 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n')
        f.write(f'# source: {db}\n')
        f.write(f'{"kernel":90s} {"calls":>8s} {"total_us":>14s} {"avg_us":>10s} {"pct":>7s}\n')
        for n, c, t, a, p in rows:
            f.write(f'{short(n):90s} {c:8d} {t:14.1f} {a:10.3f} {p:7.2f}\n')
    print(open(out).read())


def pmc(db_fetch, db_write, out):
    res = {}
    for key, db in (('FETCH_SIZE', db_fetch), ('WRITE_SIZE', db_write)):
        cur = sqlite3.connect(db).cursor()
        q = ('select kernel_name, count(*), sum(value), avg(value) from counters_collection where counter_name = ? '
             'group by kernel_name order by sum(value) desc')
        for n, c, s, a in cur.execute(q, (key,)):
            r = res.setdefault(short(n), {})
            r[key + '_launches'] = c
            r[key + '_KiB_total'] = s
            r[key + '_KiB_avg_per_launch'] = a
    from diff_sampler_amd import build
    json.dump({'meta': {'session': os.path.dirname(db_fetch), 'kernel_source_sha256': build.source_hashes(),
                        'note': 'hashes of csrc/<tu>.hip + all headers at collection time: bench.py reports a kernel\'s traffic only while '
                                'the hash of its translation unit is unchanged'},
               'note': 'raw rocprofv3 values in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x '
                       '(MI355X_MICROARCH.md, HBM section): double it before comparing with byte counts',
               'kernels': res}, open(out, 'w'), indent=1)
    for k, v in list(res.items())[:8]:
        print(k, v)


def counters(out, *dbs):
    """Generic: every counter of every pass, averaged per kernel launch -> JSON {kernel: {counter: avg, launches: n}}."""
    res = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        q = 'select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name'
        for n, cn, c, a in cur.execute(q):
            r = res.setdefault(short(n), {})
            r[cn] = a
            r['launches'] = c
    json.dump(res, open(out, 'w'), indent=1)
    for k, v in res.items():
        print(k, json.dumps(v))


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == 'counters':
        counters(sys.argv[2], *sys.argv[3:])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
