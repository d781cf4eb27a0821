#!/usr/bin/env python3
"""Condense a gpurun_out/<tag>/ profile directory (tools/gpu_round_profile.sh) into the small
text/json files kept under profiles/."""
import collections
import csv
import glob
import json
import os
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
walker = sys.argv[4] if len(sys.argv) > 4 else 'straight'       # the straight walker's counters feed bench.py's roofline line (traffic_env_step.json)
os.makedirs(dst, exist_ok=True)
out = {}
stats = glob.glob(os.path.join(src, 'bench_trace', '*', '*kernel_stats.csv'))
lines = []
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    flags = {'straight': '', 'randomize': ' --randomize', 'policy': ' --policy', 'policy_per_rollout': ' --policy --moments per_rollout', 'policy_launches': ' --policy --rollout-form launches',
             'policy_32768_h2': ' --policy --envs-per-gpu 32768 --handles 2 --steps 2', 'policy_walking': ' --policy --checkpoint walking --warmup 8',
             'policy_walking_per_rollout': ' --policy --checkpoint walking --warmup 8 --moments per_rollout'}.get(walker, f' --walker {walker}')
    lines.append('rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline' + flags + '   (kernel_stats.csv, top rows)')
    lines.append(f'{"kernel":70s} {"calls":>7s} {"avg_us":>12s} {"min_us":>10s} {"max_us":>10s} {"pct":>7s}')
    for r in rows[:14]:
        lines.append(f'{r["Name"][:70]:70s} {r["Calls"]:>7s} {float(r["AverageNs"]) / 1e3:12.2f} {float(r["MinNs"]) / 1e3:10.2f} {float(r["MaxNs"]) / 1e3:10.2f} {float(r["Percentage"]):7.2f}')
        if 'k_env_step' in r['Name'] or 'k_rollout_persistent' in r['Name'] or 'k_rollout_pairs' in r['Name']:
            out['k_env_step_avg_us'] = float(r['AverageNs']) / 1e3
            out['k_env_step_calls'] = int(r['Calls'])
pmc = {}
for p in sorted(glob.glob(os.path.join(src, 'pmc*', '*', '*counter_collection.csv'))):
    acc = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(p)):
        if 'k_env_step' in r['Kernel_Name'] or 'k_rollout_persistent' in r['Kernel_Name'] or 'k_rollout_pairs' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            meta = r
    for k, v in acc.items():
        pmc[k] = sum(v) / len(v)
    if meta:
        out['vgpr'] = meta.get('VGPR_Count'); out['agpr'] = meta.get('Accum_VGPR_Count'); out['sgpr'] = meta.get('SGPR_Count')
        out['lds_bytes'] = meta.get('LDS_Block_Size'); out['scratch_bytes_per_lane'] = meta.get('Scratch_Size')
        out['grid'] = meta.get('Grid_Size'); out['workgroup'] = meta.get('Workgroup_Size')
out['pmc_per_launch'] = pmc
if 'FETCH_SIZE' in pmc and 'WRITE_SIZE' in pmc:
    # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads
    # (MI355X_MICROARCH.md, HBM section) -> doubled for the corrected figure
    out['hbm_bytes_per_launch_raw'] = (pmc['FETCH_SIZE'] + pmc['WRITE_SIZE']) * 1024
    out['hbm_bytes_per_launch'] = (2 * pmc['FETCH_SIZE'] + pmc['WRITE_SIZE']) * 1024
lines.append('')
lines.append('PMC counters of the dominant kernel (k_env_step* / k_rollout_persistent; average per launch over the launch schedule of the benchmark (dl_rollout_fixed: one launch of 512 control steps), whole grid; separate rocprofv3 --pmc passes on tools/prof_step.py):')
for k in sorted(pmc):
    lines.append(f'  {k:24s} {pmc[k]:16.1f}')
if 'SQ_ACTIVE_INST_VALU' in pmc and 'GRBM_GUI_ACTIVE' in pmc:
    lines.append('')
    lines.append('  VALU busy / SIMD time (SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 32)): %.3f' % (pmc['SQ_ACTIVE_INST_VALU'] / (1024 * pmc['GRBM_GUI_ACTIVE'] / 32)))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in pmc:
        lines.append('  MFMA busy / SIMD time (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)): %.3f' % (pmc['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * pmc['GRBM_GUI_ACTIVE'] / 8)))
    if 'TCP_TCC_READ_REQ_sum' in pmc and 'k_env_step_avg_us' in out:
        lines.append('  L2 -> CU read requests: %.3g per launch = %.2f TB/s at 64 B per request over the %.1f ms of the launch' % (pmc['TCP_TCC_READ_REQ_sum'], pmc['TCP_TCC_READ_REQ_sum'] * 64 / out['k_env_step_avg_us'] / 1e6, out['k_env_step_avg_us'] / 1e3))
mix = None
if all(k in pmc for k in ('SQ_INSTS_VALU', 'SQ_INSTS_VALU_ADD_F32', 'SQ_INSTS_VALU_MUL_F32', 'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_TRANS_F32', 'SQ_INSTS_VALU_INT32')):
    # the DYNAMIC mix of the dominant kernel: what the hardware's per-class VALU counters say, the rest by difference.  "other" = v_mov / v_cndmask / v_readlane / v_writelane
    # / compares / max-min / DPP moves: instructions that occupy an issue slot and compute no float result (the static listing splits them further: tools/asm_mix2.py)
    tot = pmc['SQ_INSTS_VALU']
    cls = {'add_f32': pmc['SQ_INSTS_VALU_ADD_F32'], 'mul_f32': pmc['SQ_INSTS_VALU_MUL_F32'], 'fma_f32': pmc['SQ_INSTS_VALU_FMA_F32'], 'trans_f32': pmc['SQ_INSTS_VALU_TRANS_F32'],
           'int32': pmc['SQ_INSTS_VALU_INT32'], 'int64': pmc.get('SQ_INSTS_VALU_INT64', 0.0), 'cvt': pmc['SQ_INSTS_VALU_CVT'], 'f64': pmc.get('SQ_INSTS_VALU_ADD_F64', 0.0)}
    cls['other (moves, selects, lane reads / writes, compares, min / max)'] = tot - sum(cls.values())
    lines.append('')
    lines.append('DYNAMIC VALU instruction mix of the dominant kernel (per launch, wave instructions; SQ_INSTS_VALU_* classes, "other" by difference from SQ_INSTS_VALU):')
    for k, v in cls.items():
        lines.append(f'  {k:70s} {v:16.0f}  {v / tot:7.3f}')
    flop_insts = cls['add_f32'] + cls['mul_f32'] + cls['trans_f32'] + 2 * cls['fma_f32']
    mix = {'valu_insts': tot, **{k.split(' ')[0]: v for k, v in cls.items()}, 'fp32_arith_frac_of_valu': (cls['add_f32'] + cls['mul_f32'] + cls['trans_f32'] + cls['fma_f32']) / tot}
    lanes = 64.0
    cyc = pmc.get('SQ_INST_CYCLES_VALU') or pmc.get('SQ_ACTIVE_INST_VALU')          # (gfx950 has no SQ_INST_CYCLES_VALU; SQ_ACTIVE_INST_VALU counts the same quad-cycles: 1.014 per VALU instruction here)
    if 'SQ_THREAD_CYCLES_VALU' in pmc and cyc:
        # SQ_THREAD_CYCLES_VALU = VALU cycles x active threads (counter_defs.yaml, AvgNumActiveThreads): the mean number of lanes EXEC left on
        lanes = pmc['SQ_THREAD_CYCLES_VALU'] / cyc
        mix['active_lanes_mean'] = lanes
        lines.append(f'  active lanes per VALU instruction (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU): {lanes:.1f} of 64')
    if 'k_env_step_avg_us' in out:
        us = out['k_env_step_avg_us']
        # (the mean lane count is over ALL VALU instructions; 14 of a row's 16 lanes carry a dof of the straight walker and the two idle ones are mostly left on: an upper bound of the useful arithmetic)
        tf = flop_insts * lanes / (us * 1e-6) / 1e12
        mix.update({'fp32_tflops_on_active_lanes': tf, 'fp32_frac_of_157_3_tf_vector_peak': tf / 157.3})
        lines.append(f'  FP32 arithmetic: {flop_insts:.4g} flop-instructions (FMA = 2) x {lanes:.1f} active lanes / {us / 1e3:.2f} ms = {tf:.2f} TFLOP/s'
                     f' = {tf / 157.3:.3f} of the 157.3 TFLOP/s FP32 vector peak (issue slots: VALU busy above; {100 * mix["fp32_arith_frac_of_valu"]:.1f} % of the issued VALU instructions are FP32 arithmetic)')
lines.append('')
lines.append(json.dumps({k: v for k, v in out.items() if k != 'pmc_per_launch'}))
b = os.path.join(src, 'bench.json')
if os.path.exists(b):
    lines.append('')
    lines.append('bench.py (the same flags, with the CPU baselines) on the same box:')
    lines.append(open(b).read().strip())
open(os.path.join(dst, f'{tag}_summary.txt'), 'w').write('\n'.join(lines) + '\n')
if 'hbm_bytes_per_launch' in out:
    extra = {}
    if 'SQ_ACTIVE_INST_VALU' in pmc and 'SQ_WAVE_CYCLES' in pmc:
        extra = {'valu_busy_frac': pmc['SQ_ACTIVE_INST_VALU'] / pmc['SQ_WAVE_CYCLES'], 'wait_frac': pmc.get('SQ_WAIT_ANY', 0.0) / pmc['SQ_WAVE_CYCLES'],
                 'valu_insts_per_launch': pmc.get('SQ_INSTS_VALU')}
        if 'GRBM_GUI_ACTIVE' in pmc:
            # the same against the SIMDs' time instead of the waves': GRBM_GUI_ACTIVE sums the 8 XCDs' clocks over the launch, the SQ cycle counters tick
            # every 4 clocks, 1024 SIMDs -- the figure that stays meaningful when a SIMD holds two waves of which one mostly sleeps (split workgroups)
            extra['valu_busy_frac_simd'] = pmc['SQ_ACTIVE_INST_VALU'] / (1024 * pmc['GRBM_GUI_ACTIVE'] / 32)
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in pmc:      # counts shader cycles (not quad-cycles: MI355X_MICROARCH.md), summed over the SIMDs: against 1024 SIMDs x the launch's cycles
                extra['mfma_busy_frac_simd'] = pmc['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * pmc['GRBM_GUI_ACTIVE'] / 8)
                extra['mfma_insts_per_launch'] = pmc.get('SQ_INSTS_MFMA')
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_code_sha16          # the device code these counters belong to: bench.py reports them only while it matches
    # ... and the launch geometry they belong to (host-side changes -- grid, workgroup, LDS, which kernel a mode dispatches to -- move what a pass measured without moving the code object)
    launch = {'grid': int(out['grid']), 'workgroup': int(out['workgroup']), 'lds_bytes': int(out['lds_bytes'])} if out.get('grid') else None
    from drloco_amd import lib as _lib
    _lib.load()
    if mix:
        extra['valu_mix'] = mix
    json.dump({**extra, 'tag': tag, 'kernel_code_sha16': kernel_code_sha16(), 'launch': launch, 'code_object': (_lib.SELECTED or {}).get('variant'), 'hbm_bytes_per_launch': out['hbm_bytes_per_launch'], 'raw_fetch_kib': pmc['FETCH_SIZE'], 'raw_write_kib': pmc['WRITE_SIZE'],
               'source': f'profiles/{tag}_summary.txt', 'correction': 'FETCH_SIZE doubled (gfx950: 128 B requests tallied at 64 B), WRITE_SIZE as reported'},
              open(os.path.join(dst, 'traffic_env_step.json' if walker == 'straight' else f'traffic_env_step_{walker}.json'), 'w'))
print('\n'.join(lines))
