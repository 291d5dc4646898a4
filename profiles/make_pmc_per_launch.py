#!/usr/bin/env python
"""profiles/pmc_per_launch.json from a PMC summary (gpurun_out/pmc_<tag>.json written by profiles/run_pmc.sh): per bench
kernel name the HBM bytes per launch -- (2 * FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled per the gfx950 note of
/opt/skills/guides/MI355X_MICROARCH.md (rocprofv3 tallies 128-byte read requests at 64 bytes) -- and the VALU / MFMA
wave-instruction counts that bench.py turns into `roofline.traffic` / `roofline.valu_frac`.
usage: make_pmc_per_launch.py gpurun_out/pmc_<tag>.json [tag]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

NAMES = {"render_bwd": "trase::render_bwd_hw_kernel<false, false, false, false>", "render_fwd": "trase::render_fwd_mf_kernel",
         "reduce_rows": "trase::reduce_rows_kernel<44, false, 16, false>", "preprocess_fwd": "trase::preprocess_fwd_raw_kernel<32, 64>",
         "preprocess_bwd": "trase::preprocess_bwd_raw_kernel<false, 64>", "emit_pairs": "trase::emit_pairs_kernel"}


def main():
    src = json.load(open(sys.argv[1]))
    tag = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    out = {"_note": f"per launch, S4 workload, from {tag}: hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE doubled per "
                    "the gfx950 correction of MI355X_MICROARCH.md), valu_insts = SQ_INSTS_VALU, mfma_insts = SQ_INSTS_MFMA"}
    from bench import source_sha16
    out["_source_sha16"] = source_sha16()     # bench.py replays these figures only for a build of the same kernel sources
    out["_from"] = tag
    for short, full in NAMES.items():
        c = src.get(full)
        if not c:
            continue
        rec = {}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rec["hbm_bytes"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
            rec["fetch_kib"], rec["write_kib"] = round(c["FETCH_SIZE"], 1), round(c["WRITE_SIZE"], 1)
        for k, name in (("valu_insts", "SQ_INSTS_VALU"), ("mfma_insts", "SQ_INSTS_MFMA"), ("wave_quad_cycles", "SQ_WAVE_CYCLES"),
                        ("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"), ("active_valu", "SQ_ACTIVE_INST_VALU")):
            if name in c:
                rec[k] = int(c[name])
        out[short] = rec
    json.dump(out, open("profiles/pmc_per_launch.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
