"""pmc_traffic.json (what bench.py reads for roofline.traffic) from the per-kernel PMC tables of profiles/pmc_traffic_summary.py.
usage: python profiles/make_pmc_traffic.py <step per-kernel json> <full per-kernel json> <out json> [<step per-kernel json, split GEMM mode>
       [<steps the driver ran> <mean nodes per step>]]"""
import json, sys
step, full = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))


def fam(table, pred):
    n = sum(v["launches"] for k, v in table.items() if pred(k))
    b = sum(v["bytes_per_launch"] * v["launches"] for k, v in table.items() if pred(k))
    return (b / n if n else None), n


out = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (csv output, kernel trace only) over profiles/pmc_step.py (four steps of the train loop) and "
                 "profiles/pmc_probe.py spmm (cfg4); bytes/launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 averaged over the launches of "
                 "the kernel family (FETCH_SIZE x2: gfx950 counts the 128-B requests of wide coalesced reads as 64 B, MI355X_MICROARCH.md HBM "
                 "section); the counters sit on the L2's memory side, so Infinity-Cache hits are included.  Scripts: profiles/refresh.sh, "
                 "pmc_traffic_summary.py, make_pmc_traffic.py", "round": 6}
for tag, pred in (("gemm_nt", lambda k: "gemm_f32_mfma_kernel<true, true" in k), ("gemm_tn", lambda k: "gemm_f32_mfma_kernel<false, false" in k),
                  ("gemm_nn", lambda k: "gemm_f32_mfma_kernel<true, false" in k), ("batch_assemble", lambda k: "batch_assemble" in k),
                  ("spmm_csr", lambda k: k.startswith("spmm_csr_kernel"))):
    b, n = fam(step, pred)
    out[f"{tag}_bytes_per_launch"], out[f"{tag}_launches_sampled"] = b, n
if len(sys.argv) > 4:
    split = json.load(open(sys.argv[4]))
    for tag, pat in (("gemm_nt", "gemm_split_kernel<true, true"), ("gemm_tn", "gemm_split_kernel<false, false"), ("gemm_nn", "gemm_split_kernel<true, false")):
        b, n = fam(split, lambda k, pat=pat: pat in k)
        out[f"{tag}_split_bytes_per_launch"], out[f"{tag}_split_launches_sampled"] = b, n
    # round 3: the default step multiplies P3 images (planes GEMMs, csrc/gemm_p3.hip)
    # (dX with the LayerNorm-backward epilogue is its own family)
    import re
    # (the loader-wave NT kernel whose sixth template argument -- the epilogue -- is 1 or 3: LayerNorm backward)
    # (round 6: the block-major-weights kernel gemm_p3_nt_sq_kernel<TM, NL, LNB, SQ> carries the epilogue as its THIRD argument)
    def epi(k):
        m = re.search(r"gemm_p3_nt_lw_kernel<\d+[,;] \d+[,;] \d+[,;] \d+[,;] \d+[,;] (\d+)[,;>]", k) or re.search(r"gemm_p3_nt_sq_kernel<\d+[,;] \d+[,;] (\d+)[,;>]", k)
        return int(m.group(1)) if m else (0 if "gemm_p3_nt" in k else None)
    lnb = lambda k: epi(k) in (1, 3)
    lnf = lambda k: epi(k) == 4                      # the layer-0 forward on [x | cached ahn] with LayerNorm + ReLU: the DOMINANT kernel
    for tag, pred in (("gemm_nt", lambda k: "gemm_p3_nt" in k and not lnb(k) and not lnf(k)), ("gemm_tn", lambda k: "gemm_p3_tn" in k),
                      ("gemm_nt_ln_bwd", lnb), ("gemm_nt_ln_fwd", lnf)):
        b, n = fam(split, pred)
        out[f"{tag}_p3_bytes_per_launch"], out[f"{tag}_p3_launches_sampled"] = b, n
    for tag, pat in (("batch_assemble_p3", "batch_assemble"), ("ln_relu_bwd_p3", "ln_relu_bwd_vec_kernel"), ("fold_adam", "gte_fold_batch_kernel")):
        b, n = fam(split, lambda k, pat=pat: pat in k)
        out[f"{tag}_bytes_per_launch"], out[f"{tag}_launches_sampled"] = b, n
    # the WHOLE step (default mode): every launch of the step's kernel families over the steps the driver ran -- the one-time
    # launches of the set-up (feature / aggregate images: p3_from_f32, the 4-rows-per-wave aggregation into an image) are not the step's
    if len(sys.argv) > 5:
        n_steps = int(sys.argv[5])
        step_kernel = lambda k: (k.startswith(("gemm_p3_", "narrow_", "head_", "gte_fold_batch", "batch_assemble", "ln_relu_")) or
                                 (k.startswith("spmm_csr_kernel") and split[k]["launches"] >= n_steps))
        tot = sum(v["bytes_per_launch"] * v["launches"] for k, v in split.items() if step_kernel(k))
        out["step_p3_bytes_per_step"] = tot / n_steps
        out["step_p3_launches_per_step"] = sum(v["launches"] for k, v in split.items() if step_kernel(k)) / n_steps
        nps = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
        out["step_p3_nodes_per_step"] = nps or None
        out["step_p3_bytes_per_node"] = (tot / n_steps / nps) if nps else None
b, n = fam(full, lambda k: "spmm_tiled_full_kernel" in k)
out["gather_cfg4_tiled_bytes_per_launch"], out["gather_cfg4_tiled_launches_sampled"] = b, n
b, n = fam(full, lambda k: k.startswith("spmm_csr_kernel<F32, 64") or k.startswith("spmm_csr_kernel<F32; 64"))
out["kernel_names"] = {"gemm_nt": "gemm_f32_mfma_kernel<true, true, ...> (A and B K-contiguous: forward)",
                       "gemm_tn": "gemm_f32_mfma_kernel<false, false, ...> (dW, split-K)", "gemm_nn": "gemm_f32_mfma_kernel<true, false, ...> (dX)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
