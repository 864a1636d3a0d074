"""Summary of tools/gpu_pmc_irv.sh: SQ counters of k_irv_u per duration class (sum over the dispatches of the class)."""
import csv, glob, collections
classes = [("tail < 9 us", 0, 9), ("9-15 us", 9, 15), ("15-30 us", 15, 30), ("heavy >= 30 us", 30, 1e9)]
tot = {c[0]: collections.Counter() for c in classes}
n = {c[0]: 0 for c in classes}
dur = {c[0]: 0.0 for c in classes}
for i in (1, 2, 3, 4):
    for f in glob.glob(f"gpurun_out/pmcirv_{i}/**/pmc_counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            if "k_irv_u" not in r["Kernel_Name"]:
                continue
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
            for name, lo, hi in classes:
                if lo <= d < hi:
                    tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
                    if i == 1 and r["Dispatch_Id"] not in seen:
                        seen.add(r["Dispatch_Id"]); n[name] += 1; dur[name] += d
for name, _, _ in classes:
    t = tot[name]
    if not t:
        continue
    wc = max(t["SQ_WAVE_CYCLES"], 1.0)
    print(f"== {name}: {n[name]} dispatches, {dur[name]:.0f} us (durations under the counter pass)")
    print("   of wave cycles: wait_any %.2f  wait_inst_any %.2f  wait_inst_lds %.2f | active: valu %.3f salu %.3f lds %.3f vmem %.3f" % (
        t["SQ_WAIT_ANY"] / wc, t["SQ_WAIT_INST_ANY"] / wc, t["SQ_WAIT_INST_LDS"] / wc, t["SQ_ACTIVE_INST_VALU"] / wc, t["SQ_ACTIVE_INST_SCA"] / wc,
        t["SQ_ACTIVE_INST_LDS"] / wc, t["SQ_ACTIVE_INST_VMEM"] / wc))
    print("   instructions: valu %.3g salu %.3g lds %.3g vmem_rd %.3g | lds: idx_active %.3g bank_conflict %.3g (%.2f of active) addr_conflict %.3g" % (
        t["SQ_INSTS_VALU"], t["SQ_INSTS_SALU"], t["SQ_INSTS_LDS"], t["SQ_INSTS_VMEM_RD"], t["SQ_LDS_IDX_ACTIVE"], t["SQ_LDS_BANK_CONFLICT"],
        t["SQ_LDS_BANK_CONFLICT"] / max(t["SQ_LDS_IDX_ACTIVE"], 1.0), t["SQ_LDS_ADDR_CONFLICT"]))
