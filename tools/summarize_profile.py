#!/usr/bin/env python
"""Condense rocprofv3 output (gpurun_out/, scratch) into the small files kept under profiles/.

  python tools/summarize_profile.py --stats gpurun_out/prof2 --fetch gpurun_out/pmc_fetch \
         --write gpurun_out/pmc_write --tag r01 --note "bench.py --steps 10 --batch 8"

Writes profiles/<tag>_kernel_stats.csv (verbatim `--kernel-trace --stats` summary),
profiles/<tag>_pmc_traffic.csv (per-kernel FETCH_SIZE / WRITE_SIZE from the separate --pmc
passes) and profiles/<tag>_summary.json (what bench.py quotes as roofline.traffic).

Optional --mfma / --active dirs: separate --pmc passes of SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE; the
summary then carries per kernel the summed counters and mfma_busy_frac = MFMA_BUSY / ((GUI_ACTIVE / 8) * 1024 SIMDs)
(MI355X_MICROARCH.md: the counter ticks 32 cycles per v_mfma_f32_32x32x16 on the SIMD that executes it, summed
over the 256 CUs x 4 SIMDs; rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs, i.e. 8 x the elapsed
cycles -- both calibrated on the RPN 3x3 conv: 3.69e7 MFMAs x 32 = 1.18e9 expected, 1.14e9 counted; 1.0 ms at
~2 GHz = 2.0e6 cycles, 15.9e6 counted), which bench.py quotes next to its arithmetic mfma_util.  source_hashes identifies the code of every
kernel source (bench.kernel_source_hashes: comments / whitespace ignored) so a stale summary is never quoted.

HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are reported in KiB-units of
1024 B; on gfx950 FETCH_SIZE under-counts wide coalesced reads by exactly 2x, so read bytes =
2 * FETCH_SIZE * 1024 (applied to the 16-B-per-lane streaming kernels here); WRITE_SIZE is taken
as reported (uncalibrated, see the guide)."""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    n = name.replace('void ', '').replace('xdet::', '')
    return n.split('(')[0]


def pmc(dirname):
    agg = collections.defaultdict(lambda: [0, 0.0])
    f = glob.glob(os.path.join(dirname, '*counter_collection.csv'))
    if not f:
        return {}
    for r in csv.DictReader(open(f[0])):
        k = short(r['Kernel_Name'])
        agg[k][0] += 1
        agg[k][1] += float(r['Counter_Value'])
    return {k: (n, v) for k, (n, v) in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--stats', required=True)
    ap.add_argument('--fetch')
    ap.add_argument('--write')
    ap.add_argument('--mfma')
    ap.add_argument('--active')
    ap.add_argument('--tag', required=True)
    ap.add_argument('--note', default='')
    a = ap.parse_args()
    out = os.path.join(ROOT, 'profiles')
    os.makedirs(out, exist_ok=True)
    st = glob.glob(os.path.join(a.stats, '*kernel_stats.csv'))[0]
    shutil.copy(st, os.path.join(out, a.tag + '_kernel_stats.csv'))
    stats = {short(r['Name']): r for r in csv.DictReader(open(st))}
    fe = pmc(a.fetch) if a.fetch else {}
    wr = pmc(a.write) if a.write else {}
    rows = []
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, (0, 0))[1])):
        nf, vf = fe.get(k, (0, 0.0))
        nw, vw = wr.get(k, (0, 0.0))
        rd = 2.0 * vf * 1024 / max(nf, 1)
        wb = vw * 1024 / max(nw, 1)
        rows.append({'kernel': k, 'launches_in_pmc_pass': nf or nw, 'FETCH_SIZE_avg_per_launch': round(vf / max(nf, 1), 1),
                     'WRITE_SIZE_avg_per_launch': round(vw / max(nw, 1), 1), 'read_bytes_per_launch_x2corr': int(rd),
                     'write_bytes_per_launch': int(wb), 'hbm_bytes_per_launch': int(rd + wb)})
    if rows:
        with open(os.path.join(out, a.tag + '_pmc_traffic.csv'), 'w') as f:
            wtr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            wtr.writeheader()
            wtr.writerows(rows)
    sys.path.insert(0, ROOT)
    import bench
    summ = {'tag': a.tag, 'command': a.note, 'source_hash': bench.kernel_source_hash(),
            'source_hashes': bench.kernel_source_hashes(), 'kernels': {}}
    mf = pmc(a.mfma) if a.mfma else {}
    ga = pmc(a.active) if a.active else {}
    for k, r in stats.items():
        e = {'calls': int(r['Calls']), 'avg_us': round(float(r['AverageNs']) / 1e3, 2), 'pct': float(r['Percentage'])}
        for row in rows:
            if row['kernel'] == k:
                e['hbm_bytes_per_launch'] = row['hbm_bytes_per_launch']
        if k in mf and k in ga and ga[k][1] > 0:
            e['mfma_busy_cycles'] = mf[k][1]
            e['gui_active_cycles'] = ga[k][1] * mf[k][0] / max(ga[k][0], 1)      # same number of launches
            e['mfma_busy_frac'] = round(e['mfma_busy_cycles'] / (e['gui_active_cycles'] / 8.0 * 1024.0), 4)
        summ['kernels'][k] = e
    # HBM bytes per image of one forward: sum over the net's own kernels (build-time fills / copies excluded) of
    # bytes per launch (PMC pass) x launches per step (stats run: calls / (steps + warmup)) / images per step
    import re
    m_s, m_w, m_b = re.search(r'--steps (\d+)', a.note), re.search(r'--warmup (\d+)', a.note), re.search(r'--batch (\d+)', a.note)
    if m_s and m_w and m_b and rows:
        nsteps, nimg = int(m_s.group(1)) + int(m_w.group(1)), int(m_b.group(1))
        tot = sum(e.get('hbm_bytes_per_launch', 0) * e['calls'] for k, e in summ['kernels'].items()
                  if not k.startswith('__amd_rocclr'))
        summ['steps_in_stats_run'] = nsteps
        summ['images_per_step'] = nimg
        summ['hbm_bytes_per_image'] = int(tot / nsteps / nimg)
    json.dump(summ, open(os.path.join(out, a.tag + '_summary.json'), 'w'), indent=1, sort_keys=True)
    print(json.dumps(summ, indent=1)[:1500])


if __name__ == '__main__':
    main()
