"""Per-kernel SQ counter table from two rocprofv3 --pmc passes (gpurun_out/pmc1, pmc2)."""
import csv, glob, sys, collections
def load(d):
    f = glob.glob(f'{d}/**/*counter_collection.csv', recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void pcf::', '').replace('pcf::', '')[:44]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
    return {k: {c: v / len(disp[k]) for c, v in cs.items()} for k, cs in acc.items()}
a, b = load(sys.argv[1]), load(sys.argv[2])
tiles = 80000
print(f"{'kernel':44s} busy_us  valu_us  mfma_us  lds_us  mfma/busy | per tile: valu mfma lds salu")
for k in sorted(a, key=lambda k: -a[k].get('SQ_BUSY_CYCLES', 0)):
    if not any(t in k for t in ('chain', 'agg_', 'rowlin', 'gemm', 'flin', 'bn_bwd', 'bnact', 'slab', 'head_', 'tail_', 'dw_reduce', 'edge_geometry')): continue
    A, B = a[k], b.get(k, {})
    busy = A['SQ_BUSY_CYCLES'] / 32 / 2400; valu = A['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / 2400
    mfma = A['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / 2400; lds = A['SQ_ACTIVE_INST_LDS'] * 4 / 256 / 2400
    waves = max(1.0, A.get('SQ_WAVES', 0))
    print(f"{k:44s} {busy:7.1f} {valu:8.1f} {mfma:8.1f} {lds:7.1f} {mfma / busy:9.2f}  | "
          f"{B.get('SQ_INSTS_VALU', 0) / tiles:8.0f} {B.get('SQ_INSTS_MFMA', 0) / tiles:4.0f} {B.get('SQ_INSTS_LDS', 0) / tiles:4.0f} {B.get('SQ_INSTS_SALU', 0) / tiles:4.0f}")
