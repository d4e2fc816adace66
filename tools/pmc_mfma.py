"""Per-kernel matrix-pipe / LDS / occupancy figures from the rocprofv3 --pmc passes of tools/pmc_mfma.sh.

Units (MI355X_MICROARCH.md 'Per-instruction cycle constants'): SQ_BUSY_CYCLES / SQ_WAVE_CYCLES count quad-cycles per SE / wave,
SQ_VALU_MFMA_BUSY_CYCLES counts cycles (= 16 per v_mfma_f32_16x16x32_f16, summed over the SIMDs).  Derived:
  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CYCLES x active share): reported as MFMA busy cycles per SIMD / kernel cycles per
                     CU where GRBM_GUI_ACTIVE is available: busy / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)
  valu_per_mfma    = SQ_INSTS_VALU / SQ_INSTS_MFMA (SQ_INSTS_VALU includes the MFMAs on gfx9: reported as counted)
  lds_conflict_frac= SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  waves            = SQ_WAVES per launch; occupancy = SQ_WAVE_CYCLES / (SQ_BUSY_CYCLES x ...) is left as the raw pair."""
import csv, glob, json, sys, collections, os
out = sys.argv[1]
# one launch = one Dispatch_Id: rows are summed per (pass, dispatch, counter) first -- rocprofv3 may emit several rows per dispatch (per XCD /
# dimension) -- and `launches` counts distinct dispatches
per_dispatch = collections.defaultdict(float)
kname = {}
n_rows = 0
for f in glob.glob('%s/sq*/*/*counter_collection.csv' % out):
    pass_dir = f.split(os.sep + 'sq')[-1].split(os.sep)[0]
    for r in csv.DictReader(open(f)):
        key = (pass_dir, r.get('Dispatch_Id', n_rows), r['Counter_Name'])
        per_dispatch[key] += float(r['Counter_Value'])
        kname[key[:2]] = r['Kernel_Name']
        n_rows += 1
if not n_rows:
    sys.stderr.write('pmc_mfma: no counter rows under %s/sq*\n' % out)
    sys.exit(3)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for (pd, did, cname), val in per_dispatch.items():
    a = acc[kname[(pd, did)]][cname]
    a[0] += val; a[1] += 1
res = {}
WANT = ('gg_pl_kernel', 'gg_plp_kernel', 'gg_plh_kernel', 'gg_plhn_kernel', 'wgrad_pl_kernel', 'bn_bwd_apply_kernel', 'bn_reduce_kernel', 'bn_apply_kernel', 'gather_gemm', 'dw_strip', 'splitk')
for k, d in acc.items():
    if not any(w in k for w in WANT):
        continue
    name = k.split('(')[0][-70:]
    n = max(v[1] for v in d.values())
    per = {c: v[0] / max(v[1], 1) for c, v in d.items()}
    e = {'launches': n, 'per_launch': {c: round(x, 1) for c, x in sorted(per.items())}}
    gui = per.get('GRBM_GUI_ACTIVE')
    if gui and per.get('SQ_VALU_MFMA_BUSY_CYCLES') is not None:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs: kernel cycles = gui / 8; matrix-pipe capacity = cycles x 256 CUs x 4 SIMDs
        e['mfma_busy_frac'] = round(per['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui / 8.0 * 256 * 4), 4)
    if per.get('SQ_INSTS_MFMA'):
        e['valu_insts_per_mfma'] = round(per.get('SQ_INSTS_VALU', 0.0) / per['SQ_INSTS_MFMA'], 2)
        e['lds_insts_per_mfma'] = round(per.get('SQ_INSTS_LDS', 0.0) / per['SQ_INSTS_MFMA'], 2)
    if per.get('SQ_LDS_IDX_ACTIVE'):
        e['lds_bank_conflict_frac'] = round(per.get('SQ_LDS_BANK_CONFLICT', 0.0) / per['SQ_LDS_IDX_ACTIVE'], 4)
    if per.get('SQ_WAVE_CYCLES') and per.get('SQ_BUSY_CYCLES'):
        e['wave_cycles_per_busy_cycle'] = round(per['SQ_WAVE_CYCLES'] / per['SQ_BUSY_CYCLES'], 2)
    if per.get('SQ_WAVE_CYCLES') and per.get('SQ_WAIT_ANY') is not None:
        e['wait_any_frac'] = round(per['SQ_WAIT_ANY'] / per['SQ_WAVE_CYCLES'], 3)
        e['wait_inst_any_frac'] = round(per.get('SQ_WAIT_INST_ANY', 0.0) / per['SQ_WAVE_CYCLES'], 3)
        e['active_inst_any_frac'] = round(per.get('SQ_ACTIVE_INST_ANY', 0.0) / per['SQ_WAVE_CYCLES'], 3)
    res[name] = e
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
res['_meta'] = {'csrc_sha256': bench.csrc_fingerprint(),
                'command': 'rocprofv3 --pmc <8 SQ counters + GRBM_GUI_ACTIVE> / <8 SQ counters> -- python3 bench.py --steps 2 --warmup 1 (two passes, counters serialise the kernels)',
                'units': 'SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over all SIMDs; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles; GRBM_GUI_ACTIVE summed over 8 XCDs'}
print(json.dumps(res, indent=1))
