#!/usr/bin/env python3
"""Build profiles/<tag>_pmc_summary.json from rocprofv3 PMC passes over ONE bench step per arithmetic mode.

Collect on the GPU box (separate --pmc passes, the program directly after `--`, as MI355X_MICROARCH.md prescribes):

    for m in fp32 bf16x6; do
      for p in "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
               "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
        rocprofv3 --kernel-trace --pmc ${p#*:} -d gpurun_out/pmc_$m_${p%%:*} -o pmc -- \
            python3 bench.py --steps 1 --warmup 1 --math $m --no-cpu-baseline --no-bf16x6
      done
    done
    python tools/pmc_summary.py gpurun_out r02        # -> gpurun_out/r02_pmc_summary.json, then copied to profiles/

Per kernel: launches, mean duration, clock during the profiled dispatch (GRBM_GUI_ACTIVE / 8 XCDs / duration; reads high on
dispatches shorter than ~0.3 ms and, profiled kernels being serialised, is NOT the clock the chip holds in the un-profiled
step -- that one comes from in-kernel s_memtime stamps), MFMA-busy fraction
(SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 4 SIMDs * CUs)), HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE
(KiB counters; FETCH_SIZE reports half of a wide coalesced read stream on gfx950) and the derived GB/s."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_agg import aggregate

CUS = 256


def short(name):
    """Mangled kernel symbol -> 'conv_kernel<9,128,8>' style name (llvm-cxxfilt / c++filt; the raw name if neither exists)."""
    import shutil
    import subprocess
    sym = name[:-3] if name.endswith(".kd") else name
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", shutil.which("c++filt")):
        if tool and os.path.exists(tool):
            try:
                sym = subprocess.run([tool, sym], capture_output=True, text=True, timeout=10).stdout.strip() or sym
                break
            except Exception:
                pass
    sym = re.sub(r"^void ", "", sym)
    sym = sym.replace("(anonymous namespace)::", "")
    sym = re.sub(r"\(.*$", "", sym)                  # drop the argument list
    sym = re.sub(r"\b(true|false)\b", lambda m: "1" if m.group(1) == "true" else "0", sym)
    return sym.replace(" ", "")


def main(root, tag, steps=None):
    # bench.py --steps 1 --warmup 1 runs warm-up, timed and instrumented (roofline block) steps: 3, and in the fp32 mode a fourth
    # (the instrumented step once more on one stream: roofline.alone)
    steps_of = {"fp32": steps or 4, "bf16x6": steps or 3}
    out = {"_comment": __doc__.split("\n\n")[0] + " Counters: rocprofv3 --pmc, one pass per counter group; "
           "FETCH_SIZE doubled (gfx950), KiB -> bytes; clock = GRBM_GUI_ACTIVE / 8 / duration.",
           # the workload the passes ran (bench.py defaults): bench.py attaches these numbers to a run of THIS workload only
           "_commit": os.environ.get("BMC_PMC_COMMIT"),
           "_workload": json.loads(os.environ.get("BMC_PMC_WORKLOAD", '{"H": 180, "W": 240, "B": 4, "L": 9, "n_c": 128, "n_b": 5}'))}
    for mode in ("fp32", "bf16x6"):
        rows = {}
        for grp in ("sq", "fetch", "write"):
            db = os.path.join(root, "pmc_%s_%s" % (mode, grp), "pmc_results.db")
            if not os.path.exists(db):
                continue
            for r in aggregate(db):
                key = (r["kernel"], r["grid"])
                rows.setdefault(key, {}).update({k: v for k, v in r.items() if k not in ("kernel", "grid")})
        merged = {}
        for (kern, grid), r in rows.items():       # merge grids of one kernel (weighted by dispatch count)
            m = merged.setdefault(short(kern), {"launches": 0, "_w": {}})
            n = r.get("dispatches", 0)
            m["launches"] += n
            for k, v in r.items():
                if k != "dispatches":
                    a = m["_w"].setdefault(k, [0.0, 0])
                    a[0] += v * n
                    a[1] += n
        res = {}
        for kern, m in merged.items():
            w = {k: a[0] / a[1] for k, a in m["_w"].items() if a[1]}
            if "avg_us" not in w or m["launches"] < 2:
                continue
            e = {"launches_per_step": round(m["launches"] / steps_of[mode], 1), "avg_launch_us": round(w["avg_us"], 2)}
            if "GRBM_GUI_ACTIVE" in w:
                cyc = w["GRBM_GUI_ACTIVE"] / 8.0
                e["in_kernel_clock_GHz"] = round(cyc / w["avg_us"] / 1e3, 3)
                if "SQ_VALU_MFMA_BUSY_CYCLES" in w:
                    e["mfma_busy_frac"] = round(w["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 4 * CUS), 4)
                if w.get("SQ_WAVE_CYCLES"):
                    for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
                        if c in w:
                            e[c.lower() + "_frac_of_wave_cycles"] = round(w[c] / w["SQ_WAVE_CYCLES"], 4)
                if w.get("SQ_LDS_IDX_ACTIVE"):
                    e["lds_bank_conflict_frac"] = round(w.get("SQ_LDS_BANK_CONFLICT", 0.0) / w["SQ_LDS_IDX_ACTIVE"], 4)
            if "FETCH_SIZE" in w and "WRITE_SIZE" in w:
                rd, wr = 2.0 * w["FETCH_SIZE"] * 1024.0, w["WRITE_SIZE"] * 1024.0
                e["hbm_read_bytes_per_launch"] = int(rd)
                e["hbm_write_bytes_per_launch"] = int(wr)
                e["hbm_bytes_per_launch"] = int(rd + wr)
                e["hbm_GBps"] = round((rd + wr) / w["avg_us"] / 1e3, 1)
            res[kern] = e
        out[mode] = dict(sorted(res.items(), key=lambda kv: -kv[1]["launches_per_step"] * kv[1]["avg_launch_us"]))
    path = os.path.join(root, "%s_pmc_summary.json" % tag)      # on the GPU box only gpurun_out/ travels back: copy to profiles/
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, {m: len(out.get(m, {})) for m in ("fp32", "bf16x6")})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r02", int(sys.argv[3]) if len(sys.argv) > 3 else None)
