"""Turn the rocprofv3 databases of tools/profile_step.sh (under gpurun_out/<dir>/prof_{fp32,bf16,c5}) and the bench lines of
tools/collect_round.sh into the committed summaries of profiles/<round>/.

    python tools/make_profiles.py gpurun_out/<dir> profiles/r02
"""
import glob
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
HEAD = {
    "fp32": "per-kernel summary of ONE bench step (B=256, fp32, HPE_STREAMS=1, serial steps): rocprofv3 --kernel-trace + separate --pmc passes",
    "bf16": "per-kernel summary of ONE bench step (B=256, bf16 encoder + fp32 regressor / SMPL, HPE_STREAMS=1, serial steps): rocprofv3 --kernel-trace + separate --pmc passes",
    "bf16nochain": "per-kernel summary of ONE bench step (B=256, bf16 encoder with HPE_CHAIN=0: one launch per layer, the round-3 plan; HPE_STREAMS=1, serial steps): rocprofv3 --kernel-trace + separate --pmc passes",
    "c5": "per-kernel summary of ONE bench step of config 5 (B=256, fp32 + kp / mesh reprojection losses of the 3 stages, HPE_STREAMS=1): rocprofv3 --kernel-trace + separate --pmc passes",
}


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    for key in ("fp32", "bf16", "bf16nochain", "c5"):
        d = os.path.join(src, "prof_" + key)
        if not os.path.isdir(d):
            continue
        dbs = [os.path.join(d, x) for x in ("kt/kt_results.db", "fetch/f_results.db", "write/w_results.db", "mfma/m_results.db")]
        out = subprocess.run([sys.executable, os.path.join(HERE, "pmc_summary.py")] + dbs, capture_output=True, text=True, check=True).stdout
        body, js = out.split("\nJSON ")
        name = {"fp32": "fp32", "bf16": "bf16", "bf16nochain": "bf16_chain_off", "c5": "config5"}[key]
        with open(os.path.join(dst, "%s_pmc_summary.md" % name), "w") as f:
            f.write("# %s\n# (tools/profile_step.sh, tools/pmc_summary.py); HBM bytes = FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE; "
                    "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)\n\n" % HEAD[key])
            f.write(body.strip() + "\n")
        if key != "c5":
            j = json.loads(js)
            j.update(batch=256, dtype=key.replace("nochain", ""), source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over the conv launches of one step")
            with open(os.path.join(dst, "final_conv_hbm_traffic_%s.json" % name), "w") as f:
                json.dump(j, f)
                f.write("\n")
        for kt, tag in (("kt/kt_results.db", "streams1"), ("kt3/kt3_results.db", "default")):
            csv = subprocess.run([sys.executable, os.path.join(HERE, "kernel_stats.py"), os.path.join(d, kt)], capture_output=True, text=True, check=True).stdout
            with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (name, tag)), "w") as f:
                f.write(csv)
    for f in glob.glob(os.path.join(src, "bench_*.json")) + glob.glob(os.path.join(src, "layers_*.txt")) + glob.glob(os.path.join(src, "power_trace_*.csv")) + \
            glob.glob(os.path.join(src, "latency_*.txt")) + glob.glob(os.path.join(src, "mesh_loss_search.txt")):
        shutil.copy(f, os.path.join(dst, os.path.basename(f)))
    kt3 = os.path.join(src, "prof_fp32", "kt3", "kt3_results.db")
    if os.path.isfile(kt3):
        out = subprocess.run([sys.executable, os.path.join(HERE, "timeline_gaps.py"), kt3], capture_output=True, text=True, check=True).stdout
        with open(os.path.join(dst, "timeline_fp32.txt"), "w") as f:
            f.write("# tools/timeline_gaps.py on the default fp32 run (2 chunk streams + pipelined tail, B = 256) of tools/profile_step.sh\n" + out)
    print("\n".join(sorted(os.listdir(dst))))


if __name__ == "__main__":
    main()
