"""Group a rocprofv3 kernel_stats.csv into families; print ms/step."""
import csv, glob, sys, collections
d, steps = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0])))
fam = collections.defaultdict(lambda: [0.0, 0])
def family(n):
    # unit-tensor families first (their names contain none of the NCHW family substrings, or shadow them)
    if "conv_small" in n: return "tbg conv_small (small maps from unit tensors, K split in the block)"
    if "conv_wgrad_units" in n: return "tbg conv_wgrad_units (filter gradient from unit tensors)"
    if "conv_units" in n or "conv_group" in n: return "tbg conv_units (fprop / dgrad / s2 / t2 from unit tensors)"
    if "units_pack" in n or "fir_units" in n or "bias_act_bwd_units" in n or "units_" in n: return "tbg unit-tensor producers"
    if "conv_fprop" in n or "conv_tmerge" in n: return "tbg conv_fprop (NCHW operands)"
    if "conv_wgrad" in n: return "tbg conv_wgrad(+reduce)"
    if "upfirdn" in n: return "tbg upfirdn2d"
    if "bias_act" in n: return "tbg bias_act"
    if "rgb_" in n: return "tbg rgb"
    if "slab_epilogue" in n: return "tbg split-K epilogue"
    if "lstm_step" in n or "lstm_fused" in n or "attn_ctx" in n or "dec_" in n: return "tbg OCR recurrent (lstm_step / attn_ctx)"
    if "dense_" in n or "smalls" in n or "mbstd" in n: return "tbg small-tensor kernels (dense / tails / mbstd)"
    if "weight_pack" in n or "weight_transpose" in n or "demod" in n or "wsq" in n or "adam" in n or "ema_" in n or "axpby" in n: return "tbg misc (pack/demod/adam/ema)"
    if n.startswith("Cijk") or "gemm" in n.lower(): return "rocBLAS/hipBLASLt GEMM"
    if "LSTM" in n or "miopen" in n.lower() or "MIOpen" in n or "Im2d" in n or "Col2Im" in n or "batched_transpose" in n or "SubTensor" in n or "naive_conv" in n or "ck::" in n: return "MIOpen (LSTM etc.)"
    if "FillFunctor" in n: return "torch fill"
    if "reduce_kernel" in n: return "torch reduce"
    if "copy" in n.lower(): return "torch/hip copy"
    if "at::native" in n: return "torch elementwise/other"
    return "other"
for r in rows:
    f = fam[family(r["Name"])]; f[0] += float(r["TotalDurationNs"]) / 1e6; f[1] += int(r["Calls"])
tot = sum(v[0] for v in fam.values())
print(f"total {tot/steps:.2f} ms/step, {sum(v[1] for v in fam.values())/steps:.0f} launches/step")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:40s} {v[0]/steps:7.2f} ms/step  {v[1]/steps:7.0f} launches/step  avg {1e3*v[0]/v[1]:6.1f} us")
