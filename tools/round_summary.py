#!/usr/bin/env python3
"""One page of numbers for profiles/rNN_summary.md, written from the files tools/make_profiles.sh has just produced (no hand-typed
figures): the bench line, the per-family kernel times of the rocprofv3 trace, the counter-derived figures.
Usage: round_summary.py NN <dir with rNN_* files>"""
import json
import os
import re
import sys

nn, d = sys.argv[1], sys.argv[2]


def rd(name):
    p = os.path.join(d, name)
    return open(p).read() if os.path.exists(p) else ""


def jline(name):
    for l in reversed(rd(name).splitlines()):
        if l.startswith("{"):
            return json.loads(l)
    return None


b = jline(f"r{nn}_bench_n1.json")
print(f"# Round {nn}: numbers of `tools/make_profiles.sh {nn}` (one MI355X, one process; written by tools/round_summary.py)\n")
if b:
    r, c = b["roofline"], b.get("cpu_baseline") or {}
    print("| | |\n|---|---|")
    print(f"| `value` (N = 1, {b['steps']} timed steps) | **{b['value']} denoise-steps/s, {b['ms_per_step']} ms per step**, {b['launches_per_step']} launches per step |")
    print(f"| `roofline` ({r['kernel']}) | {r['achieved']} TFLOP/s = **{r['frac']}** of the f32-MFMA peak; {r['launches_per_step']} launches, {r['avg_launch_us']} us average; "
          f"counter-side {r.get('hbm_counter_GBs')} GB/s, matrix pipe busy {r.get('mfma_util_counter')} |")
    ra = b.get("roofline_attention") or {}
    print(f"| `roofline_attention` (k_attention) | {ra.get('achieved')} TFLOP/s = **{ra.get('frac')}**; {ra.get('launches_per_step')} launches; matrix pipe busy {ra.get('mfma_util_counter')} |")
    for k, v in (b.get("families") or {}).items():
        print(f"| family `{k}` | {v['ms_per_step']} ms per step, {v['launches']} launches" + (f", {v['tflops']} TFLOP/s" if v.get("tflops") else "") +
              (f", {v['algorithmic_GBs']} GB/s algorithmic" if v.get("algorithmic_GBs") else "") + " |")
    if c:
        print(f"| `cpu_baseline` | {c['value']} steps/s at {c['cores']} threads ({c['sample'][:70]}...); probe s/step {c.get('thread_probe_s_per_step')}; leg wall {c.get('leg_wall_s')} s |")
    print(f"| `gpu_active_s` | {b.get('gpu_active_s')} |")
    hv = (r.get("hbm_view") or {}).get("deep_levels")
    if hv:
        ctr = f"; counter side {hv['counter_GBs']} GB/s = {hv['counter_over_algorithmic']} x the algorithmic bytes" if hv.get("counter_GBs") else ""
        print(f"| `roofline.hbm_view.deep_levels` | {hv['achieved_GBs']} GB/s algorithmic = **{hv['frac']}** of 8 TB/s over {hv['launches']} k_deep_conv launches ({hv['ms_per_step']} ms per step){ctr} |")
    r6 = b.get("res64_info")
    if r6:
        print(f"| `res64_info` (configs[3], informational) | {r6['steps_per_s']} steps/s, {r6['ms_per_step']} ms per step; conv {r6['conv_frac_of_f32_mfma_peak']} / attention {r6['attention_frac_of_f32_mfma_peak']} of the f32-MFMA peak |")
    bi, ai = b.get("batched_info"), b.get("autoencoder_info")
    if bi:
        print(f"| 8 clips batched (informational) | {bi['clip_steps_per_s']} clip-steps/s, conv {bi['k_conv_frac_of_f32_mfma_peak']} / attention {bi['k_attention_frac_of_f32_mfma_peak']} of the f32-MFMA peak |")
    if ai:
        print(f"| autoencoder (informational) | decode {ai['decode_from_sample_ms']} ms, extract {ai['extract_ms']} ms, GEMMs {ai.get('gemm_frac_of_f32_mfma_peak')} of peak; clip end to end {ai['clip_end_to_end_ms_at_250_steps']} ms |")
b64 = jline(f"r{nn}_bench_res64_n1.json")
if b64:
    c = b64.get("cpu_baseline") or {}
    print(f"| configs[3] (R = 64) | {b64['value']} steps/s, {b64['ms_per_step']} ms per step; CPU oracle {c.get('value')} steps/s at {c.get('cores')} threads ({(c.get('sample') or '')[:40]}...) |")
print()
po = rd(f"r{nn}_per_op_rocprof.txt")
if po:
    print("## Kernel time per family (rocprofv3 kernel trace of the hipGraph replay, median per launch)\n\n```")
    print("\n".join(l for l in po.splitlines() if l.startswith("#")))
    print("```\n")
ss = rd(f"r{nn}_step_summary.txt")
if ss:
    print("## Per-step kernel view and the PMC bytes of the same passes\n\n```")
    print("\n".join(ss.splitlines()[:40]))
    print("```\n")
pu = rd(f"r{nn}_pmc_util.txt")
if pu:
    print("## Counter-derived utilisation (tools/pmc_util.py; PMC passes: " + " ".join((rd("pmc_modes.txt") or rd(f"r{nn}_pmc_modes.txt") or "").split()) + ")\n\n```")
    print(pu.strip())
    print("```\n")
dc = rd(f"r{nn}_deep_chain.txt")
if dc:
    print("## Deep-level kernels on their own (tools/ubench/deep_bench)\n\n```")
    print("\n".join(l for l in dc.splitlines() if not l.lstrip().startswith("stamps")))
    print("```")
for name, title in ((f"r{nn}_conv_win_chain.txt", "k_conv tiles vs k_conv_win tiles, 20 dependent 3x3 convs in one graph (deep_bench win; tile = MT,NT,NW,KS,XM, NW 80 = k_conv_win)"),
                    (f"r{nn}_conv_pw_chain.txt", "k_conv tiles vs k_conv_pw tiles, 20 launches of a qkv-shaped 1x1 conv in one graph (deep_bench pw; NW 96 = k_conv_pw)")):
    t = rd(name)
    if t:
        print(f"\n## {title}\n\n```")
        print("\n".join(l for l in t.splitlines() if not l.lstrip().startswith("stamps") and not l.startswith("+ ")))
        print("```")

db = rd(f"r{nn}_deep_block.txt")
if db:
    print("\n## k_deep_block: a deep-level attention block in one launch (deep_bench block time: vs the double-precision CPU block; graph chain of 20 launches; stamps = shader-clock ticks since entry)\n\n```")
    print("\n".join(l for l in db.splitlines() if l.startswith("block") or "us per launch" in l or "stamps wg 0" in l or "CHECK" in l or "differ" in l))
    print("```")
for name, title in ((f"r{nn}_deep_block_ab.txt", "same-box A/B: three launches per deep attention block vs k_deep_block where the rule selects it"),
                    (f"r{nn}_deep_block_all_ab.txt", "same-box A/B: the rule vs k_deep_block on every deep attention block"),
                    (f"r{nn}_tagged_fin_ab.txt", "same-box A/B: k_deep_finalize passes vs completion inside the producing launch (tagged granules)"),
                    (f"r{nn}_pool_fold_ab.txt", "same-box A/B: k_pool_down launches vs AvgPool2d inside k_deep_conv"),
                    (f"r{nn}_steps_per_graph_ab.txt", "sampler steps per hipGraph")):
    t = rd(name)
    if t:
        print(f"\n## {title}\n\n```")
        print(t.strip())
        print("```")
