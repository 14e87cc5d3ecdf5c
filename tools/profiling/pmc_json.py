# usage: pmc_json.py <fetch_dir> <write_dir> <out.json>  -- HBM traffic per dispatch of every pcf:: kernel of the layer bench
import csv, glob, sys, json, collections, re
def load(d, name):
    f = glob.glob(f'{d}/**/*counter_collection.csv', recursive=True)[0]
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != name: continue
        k = r['Kernel_Name'].split('(')[0]
        k = re.sub(r'^void ', '', k).replace('pcf::', '')
        tot[k] += float(r['Counter_Value']); cnt[k] += 1
    return tot, cnt
fe, fc = load(sys.argv[1], 'FETCH_SIZE')
wr, wc = load(sys.argv[2], 'WRITE_SIZE')
N, K = 80000, 16
alg = {'agg_bwd_fx_mfma_kernel': 501760000, 'agg_fwd_fx_mfma_kernel': 296960000,
       'tail_fwd_kernel<16, 2, 4, 1>': N * (256 + 32) * 4, 'tail_fwd_kernel<16, 2, 4, 2>': N * (32 + 64) * 4,
       'tail_bwd_kernel<16, 2, 4, 1>': N * 64 * 4 * 4, 'tail_bwd_kernel<16, 2, 4, 2>': N * (64 + 64 + 32 + 32) * 4,
       'tail_bwd_kernel<16, 2, 4, 3>': N * (32 + 32 + 256) * 4, 'head_fwd_kernel<4, 1, 2, 1>': N * (64 + 16) * 4,
       'head_fwd_kernel<4, 1, 2, 2>': N * (16 + 16) * 4, 'head_fwd_kernel<4, 1, 2, 3>': N * (16 + 8) * 4,
       'head_bwd_kernel<4, 1, 2, 1>': N * (8 + 16) * 4, 'head_bwd_kernel<4, 1, 2, 2>': N * (8 + 16 * 4) * 4,
       'head_bwd_kernel<4, 1, 2, 3>': N * (16 + 16 + 64 + 64) * 4, 'flin_bwd_w_kernel<1, 4, true>': N * (32 + 32 + 256) * 4}
out = {'_method': 'rocprofv3 --pmc FETCH_SIZE (pass 1) and --pmc WRITE_SIZE (pass 2), each with --kernel-trace only, on '
                  '`python3 bench.py --no-cpu-baseline --no-graph --steps 3 --warmup 2` (MI355X; scratch/profiles_r02.sh). Counters are '
                  'in KiB per dispatch, mean over the dispatches of the run. FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM '
                  'section) prescribes for gfx950 wide coalesced reads; WRITE_SIZE is taken as it is.',
       'shape': {'N': N, 'K': K, 'Ci': 16, 'Cm': 16, 'H': 8}, 'kernels': {}}
for k in sorted(fe):
    if 'at::' in k or 'rocclr' in k or 'knn' in k or 'cell_' in k or 'scan_' in k or 'csr' in k or 'seg_grid' in k or 'rocprim' in k: continue
    f = fe[k] / max(1, fc[k]); w = wr.get(k, 0.0) / max(1, wc.get(k, 1))
    e = {'FETCH_SIZE_KiB': round(f, 1), 'WRITE_SIZE_KiB': round(w, 1), 'traffic_bytes': int(2 * f * 1024 + w * 1024), 'dispatches': fc[k]}
    if k in alg:
        e['algorithmic_bytes'] = alg[k]; e['traffic_over_algorithmic'] = round(e['traffic_bytes'] / alg[k], 3)
    out['kernels'][k] = e
json.dump(out, open(sys.argv[3], 'w'), indent=1)
for k, e in out['kernels'].items():
    if 'algorithmic_bytes' in e: print(k, e['traffic_bytes'], e['algorithmic_bytes'], e['traffic_over_algorithmic'])
