"""Every heavy kernel of the headline step against its OWN roof: algorithmic bytes and FLOPs of the launch (shape formulas below),
bound = max(bytes / 5.9 TB/s, FLOPs / 1.9 PFLOP/s) -- the rates this chip sustains under this load (the best streaming kernel of
the step; the MFMA rate at the 1.84 GHz it clocks) -- against the exclusive average of
profiles/r03_step_kernel_stats_exclusive.csv.  Prints the markdown table of DESIGN.md section 5a.
    python tools/gap_table.py [csv] [steps in the profile]"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_step_kernel_stats_exclusive.csv")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
HBM, MFMA = 5.9e12, 1.9e15
Ms, Mt, Ml, D = 44160, 25216, 18944, 384          # student tokens, teacher (= global-crop) tokens, local-crop tokens, width
# name fragment -> (what, bytes per launch, flops per launch); launches that alternate two shapes carry the mean
K = {
    "panel_kernel<11, 8, 64, true, 1, 0, 1>": ("fc1 / qkv dX + LayerNorm backward", Ms * (2 * 1344 + 5376), 2 * Ms * D * 1344),
    "dw8_group_kernel": ("weight gradients of a block", 557e6, 2 * Ms * (D * 1152 + D * D + 2 * D * 1536)),
    "panel_kernel<11, 8, 64, false, 0, 0, 1>": ("proj / fc2 + LayerNorm forward, student", Ms * (2 * 960 + 3840), 2 * Ms * D * 960),
    "panel_kernel<11, 8, 64, false, 2, 3, 1>": ("fc1 + GELU, pre-activation saved", Ms * (2 * D + 4 * 1536), 2 * Ms * D * 1536),
    "panel_kernel<7, 8, 64, false, 0, 0, 1>": ("proj / fc2 + LayerNorm forward, teacher", Mt * (2 * 960 + 3840), 2 * Mt * D * 960),
    "panel_kernel<11, 8, 64, true, 2, 4, 1>": ("GELU' . dX of fc2", Ms * (2 * D + 4 * 1536), 2 * Ms * D * 1536),
    "attn_bwd_kernel<14, 2>": ("attention backward, 197 tokens", Mt * 6144, 10 * 197 * 197 * 64 * 768),
    "panel_kernel<12, 8, 64, false, 2, 1, 1>": ("qkv, student", Ms * (2 * D + 2 * 1152), 2 * Ms * D * 1152),
    "panel_kernel<7, 8, 64, false, 2, 2, 1>": ("fc1 + GELU, teacher", Mt * (2 * D + 2 * 1536), 2 * Mt * D * 1536),
    "attn_fwd_varlen_kernel": ("attention forward, student (197 + 37 tokens, one launch)", Ms * 3072, 4 * 64 * (197 * 197 * 768 + 37 * 37 * 3072)),
    "panel_kernel<7, 8, 64, false, 2, 1, 1>": ("qkv, teacher", Mt * (2 * D + 2 * 1152), 2 * Mt * D * 1152),
    "attn_bwd_kernel<4, 1>": ("attention backward, 37 tokens", Ml * 6144, 10 * 37 * 37 * 64 * 3072),
    "attn_fwd_kernel<14>": ("attention forward, teacher", Mt * 3072, 4 * 64 * 197 * 197 * 768),
    "adamw_ema_kernel": ("AdamW + teacher EMA + 16-bit refresh", 40 * 44.0e6 / 2, 0),
    "panel_kernel<11, 8, 64, true, 2, 0, 1>": ("proj dX", Ms * 4 * D, 2 * Ms * D * D),
}
rows = list(csv.DictReader(open(path)))
total = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e3
print("| kernel | launches / step | MB, GFLOP per launch | roof (µs) | measured (µs) | above the roof, per step (µs) |")
print("|---|---|---|---|---|---|")
gap_sum = cov = 0.0
for frag, (what, by, fl) in K.items():
    r = next(r for r in rows if frag in r["Name"])
    n, us = int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3
    t_b, t_f = by / HBM * 1e6, fl / MFMA * 1e6
    roof = max(t_b, t_f)
    gap = (us - roof) * n
    gap_sum += gap; cov += us * n
    print(f"| {what} (`{frag}`) | {n:g} | {by / 1e6:.0f}, {fl / 1e9:.1f} | {roof:.0f} ({'HBM' if t_b >= t_f else 'MFMA'}) | {us:.1f} | {gap:.0f} |")
print(f"| **sum of the rows** ({cov / 1e3:.2f} of the {total / 1e3:.2f} ms of kernels per step) | | | | | **{gap_sum / 1e3:.2f} ms** |")
