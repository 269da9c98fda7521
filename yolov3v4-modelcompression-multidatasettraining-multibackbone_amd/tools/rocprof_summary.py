#!/usr/bin/env python3
"""Text summaries of rocprofv3 result databases (rocpd sqlite): kernel stats and PMC sums per kernel.

    python tools/rocprof_summary.py stats  <results.db>              # like --stats: calls, total, avg, %
    python tools/rocprof_summary.py pmc    <results.db> [<more.db>]  # per kernel: sum / per-dispatch mean of every counter
    python tools/rocprof_summary.py list   <results.db> [<more.db>]  # every dispatch in order with its counters
    python tools/rocprof_summary.py traffic <out.json> <FETCH_SIZE.db> <WRITE_SIZE.db>   # HBM bytes per dispatch per kernel
"""
import sqlite3
import sys


def short(name):
    import re
    ty = {'DF16_': 'f16', 'f': 'f32', 'a': 'i8'}
    m = re.match(r'_ZN2yh2[0-9]conv_igemm_(pp2?|k64)_kernelI(DF16_|f|a)(DF16_|f|a)Li(\d+)ELi(\d+)E', name)
    if m:   # wave grid WM x WN of 128 x 64 (f16) / 128 x 64 (int8) wave tiles: 2x4 = 256 x 256, 4x2 = 512 x 128, 1x8 = 128 x 512
        return 'conv_igemm_%s<%s,%s,%dx%d>' % (m.group(1), ty[m.group(2)], ty[m.group(3)], 128 * int(m.group(4)), 64 * int(m.group(5)))
    m = re.match(r'_ZN2yh22conv_igemm_glds_kernelIaaLi(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d+)ELi(\d+)E', name)
    if m:
        return 'conv_igemm_glds<i8,i8,%sx%s,S%s>' % (m.group(1), m.group(2), m.group(3))
    # rocprofv3 demangles the int8 instantiations itself ("void yh::conv_igemm_glds_kernel<signed char, signed char, 256, 128, ...")
    dty = {'signed char': 'i8', 'float': 'f32', '_Float16': 'f16'}
    m = re.match(r'void yh::conv_igemm_glds_kernel<(signed char|float|_Float16), (signed char|float|_Float16), (\d+), (\d+), \d+, \d+, (\d+), (\d+)>', name)
    if m:
        return 'conv_igemm_glds<%s,%s,%sx%s,S%s>' % (dty[m.group(1)], dty[m.group(2)], m.group(3), m.group(4), m.group(5))
    m = re.match(r'void yh::conv_igemm_(pp2?|k64)_kernel<(signed char|float|_Float16), (signed char|float|_Float16), (\d+), (\d+)>', name)
    if m:
        return 'conv_igemm_%s<%s,%s,%dx%d>' % (m.group(1), dty[m.group(2)], dty[m.group(3)], 128 * int(m.group(4)), 64 * int(m.group(5)))
    m = re.match(r'_ZN2yh17conv_igemm_kernelI(DF16_|f)(DF16_|f)Li(\d+)ELi(\d+)E', name)
    if m:
        return 'conv_igemm<%s,%s,%sx%s>' % ('f16' if m.group(1) != 'f' else 'f32', 'f16' if m.group(2) != 'f' else 'f32',
                                           m.group(3), m.group(4))
    m = re.match(r'_ZN2yh22conv_igemm_glds_kernelI(DF16_|f)(DF16_|f)Li(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d+)ELi(\d+)E', name)
    if m:
        return 'conv_igemm_glds<%s,%s,%sx%s,S%s%s>' % ('f16' if m.group(1) != 'f' else 'f32', 'f16' if m.group(2) != 'f' else 'f32',
                                                       m.group(3), m.group(4), m.group(5), '' if m.group(6) == '0' else ',abl' + m.group(6))
    m = re.match(r'_ZN2yh19conv3x3_halo_kernelI(DF16_|f)(DF16_|f)Li(\d+)E', name)
    if m:
        return 'conv3x3_halo<%s,%s,%sx256>' % ('f16' if m.group(1) != 'f' else 'f32', 'f16' if m.group(2) != 'f' else 'f32', m.group(3))
    # halo ping-pong conv (all halo-piece counts / roles of one precision together) and the weight-gradient kernels
    m = re.match(r'(?:void yh::conv3x3_hpp_kernel<(signed char|_Float16),|_ZN2yh18conv3x3_hpp_kernelI(DF16_|a))', name)
    if m:
        t = 'i8' if (m.group(1) == 'signed char' or m.group(2) == 'a') else 'f16'
        return 'conv3x3_hpp<%s,%s,128x512>' % (t, t)
    m = re.match(r'(?:void yh::conv3x3_stream_kernel<(signed char|_Float16),|_ZN2yh21conv3x3_stream_kernelI(DF16_|a))', name)
    if m:
        return 'conv3x3_stream<%s>' % ('i8' if (m.group(1) == 'signed char' or m.group(2) == 'a') else 'f16')
    if 'conv_wgrad_roll_kernel' in name:
        return 'conv_wgrad_roll'
    if 'wgrad_roll_reduce' in name:
        return 'wgrad_roll_reduce'
    if 'conv_wgrad_halo_kernel' in name:
        return 'conv_wgrad_halo'
    if 'wgrad_halo_reduce_kernel' in name:
        return 'wgrad_halo_reduce'
    m = re.match(r'(?:void yh::conv_wgrad_dma_kernel<(\d+), (\d+)|_ZN2yh21conv_wgrad_dma_kernelILi(\d+)ELi(\d+))', name)
    if m:
        return 'conv_wgrad_dma<%s,%s>' % (m.group(1) or m.group(3), m.group(2) or m.group(4))
    m = re.match(r'_ZN2yh(\d+)([a-z_0-9]+)', name)
    if m:
        return m.group(2)[:int(m.group(1))]
    return name.split('(')[0][:60]


def stats(db):
    c = sqlite3.connect(db)
    rows = list(c.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    print('%-44s %8s %14s %12s %7s' % ('kernel', 'calls', 'total_ms', 'avg_ms', '%'))
    for name, calls, total, avg, pct in rows:
        print('%-44s %8d %14.3f %12.5f %6.2f%%' % (short(name), calls, total / 1e3, avg / 1e3, pct))


def pmc(dbs):
    agg = {}
    for db in dbs:
        c = sqlite3.connect(db)
        q = 'select kernel_name, counter_name, sum(value), count(*), sum(duration) from counters_collection group by kernel_name, counter_name'
        for k, cn, s, n, dur in c.execute(q):
            agg.setdefault(short(k), {})[cn] = (s, n, dur)
    names = sorted({cn for v in agg.values() for cn in v})
    for k, v in sorted(agg.items(), key=lambda kv: -max(x[2] for x in kv[1].values())):
        n = max(x[1] for x in v.values())
        print('%s  (dispatches %d)' % (k, n))
        for cn in names:
            if cn in v:
                s, cnt, dur = v[cn]
                print('    %-28s sum %.6g   per-dispatch %.6g   (profiled us/dispatch %.2f)' % (cn, s, s / cnt, dur / cnt / 1e3))


def per_dispatch(db, counters):
    """{dispatch_id: (short kernel name, grid, duration ns, {counter: value})}: a counter's rows of one dispatch are SUMMED (rocprofv3
    may store one row per counter instance), dispatches are never averaged across kernels of different sizes here."""
    c = sqlite3.connect(db)
    rows = {}
    q = 'select dispatch_id, kernel_name, grid_size, counter_name, value, duration from counters_collection'
    for did, k, g, cn, v, dur in c.execute(q):
        if counters and cn not in counters:
            continue
        r = rows.setdefault(did, [short(k), g, dur, {}])
        r[3][cn] = r[3].get(cn, 0.0) + v
    return rows


def listing(dbs):
    """Every dispatch in order: kernel, grid, duration, counters (the form the r04 calibration and per-layer traffic tables use)."""
    for db in dbs:
        print('==', db)
        rows = per_dispatch(db, None)
        for did in sorted(rows):
            k, g, dur, cs = rows[did]
            print('%5d %-44s grid %9d  %8.1f us  %s' % (did, k[:44], g, dur / 1e3, '  '.join('%s=%.6g' % kv for kv in sorted(cs.items()))))


def traffic(dbs, out_path):
    """HBM bytes per dispatch per kernel from separate FETCH_SIZE / WRITE_SIZE passes.

    Units: both counters are KiB.  Calibration on this part (profiles/r04_traffic_calibration.txt: single launches with known
    byte counts): WRITE_SIZE is exact (all requests 64 B); FETCH_SIZE = 64 B x TCC_EA0_RDREQ, i.e. a 128-byte request is
    tallied at 64 bytes (wide coalesced streams read exactly half: MI355X_MICROARCH.md, HBM) while a 64-byte request (a
    channel-slice row, an LDS-DMA row piece) is tallied in full.  `hbm_bytes_per_dispatch` applies the guide's doubling to the
    read side - exact for streaming kernels, an upper bound for kernels that fetch 64-byte pieces; `fetch_kib_raw` is kept.
    Means are taken over DISPATCHES of a kernel name (round 3 averaged rows of a merged-name group, which mis-stated
    conv3x3_hpp's write side by 2.4x; per dispatch its WRITE_SIZE equals its output bytes exactly)."""
    import json
    agg = {}
    for db in dbs:
        for did, (k, g, dur, cs) in per_dispatch(db, ('FETCH_SIZE', 'WRITE_SIZE')).items():
            for cn, v in cs.items():
                a = agg.setdefault(k, {}).setdefault(cn, [0.0, 0])
                a[0] += v
                a[1] += 1
    res = {}
    for k, v in agg.items():
        if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            f, w = v['FETCH_SIZE'][0] / v['FETCH_SIZE'][1], v['WRITE_SIZE'][0] / v['WRITE_SIZE'][1]
            res[k] = dict(fetch_kib_raw=round(f, 1), write_kib=round(w, 1), dispatches=v['WRITE_SIZE'][1],
                          hbm_bytes_per_dispatch=int((2 * f + w) * 1024))
    json.dump(res, open(out_path, 'w'), indent=1, sort_keys=True)
    print('wrote', out_path, len(res), 'kernels')


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2])
    elif sys.argv[1] == 'traffic':
        traffic(sys.argv[3:], sys.argv[2])
    elif sys.argv[1] == 'list':
        listing(sys.argv[2:])
    else:
        pmc(sys.argv[2:])
