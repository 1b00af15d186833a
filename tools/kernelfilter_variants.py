"""KernelFilter block (512->32 with the folded F1 + LeakyReLU, 32->512 with the folded F2 + residual; x 3 per frame) under the
split-K variants of the 512->32 convolution (RRV_KSPLIT = 1 / 2 / 4 / 8; the library's rule picks 8 at 80 x 80, 4 at
144 x 144): per-frame time of its kernels and the frame rate, headline configuration and config 5.  (run on the GPU box)
    python tools/kernelfilter_variants.py > profiles/r05_kernelfilter.txt"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KF = ("conv_wino<0>", "sum_parts", "conv_wino<E_LRELU>", "conv_wino<E_RES>", "conv_wino<E_RES | E_NORM2>")
for cfg, extra in (("headline: 640 x 640, sixteen frames per launch", []), ("config 5: 1152 x 1152, four styles, one frame per launch", ["--multistyle", "4", "--size", "1024"])):
    print("# " + cfg)
    for split in ("rule", "1", "2", "4", "8"):
        env = dict(os.environ)
        env.pop("RRV_KSPLIT", None)
        if split != "rule":
            env["RRV_KSPLIT"] = split
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", "8", "--warmup", "2"] + extra,
                           env=env, capture_output=True, text=True, timeout=1200)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            print("split %s: bench failed: %s" % (split, (r.stdout + r.stderr)[-400:]))
            continue
        rows = {k["kernel"]: k for k in d["kernels"]}
        parts = ["%s %.4f ms (%.2f of the fp32-MFMA peak)" % (n, rows[n]["ms_per_frame"], rows[n].get("frac_of_mfma_peak", 0.0)) if "frac_of_mfma_peak" in rows[n]
                 else "%s %.4f ms" % (n, rows[n]["ms_per_frame"]) for n in KF if n in rows]
        block = sum(rows[n]["ms_per_frame"] for n in KF if n in rows)
        print("RRV_KSPLIT=%-4s %.1f frames/s | KernelFilter block %.4f ms per frame | %s" % (split, d["value"], block, " | ".join(parts)), flush=True)
