"""profiles/pmc_valu.json from an SQ counter pass of bench.py (rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS): vector instructions per wave of the dominant kernel and where its
wave-cycles go.  bench.py multiplies valu_per_wave by the waves of a step for roofline.achieved; the file is only trusted when its
workload_key and source_key (hash of csrc/) match the run.   usage: pmc_valu.py <pmc dir> <kernel substring> <workload_key> <out.json>"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_key

f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
sub = sys.argv[2]
acc = {}; disp = set()
for r in csv.DictReader(open(f)):
    if sub not in r["Kernel_Name"]:
        continue
    acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    disp.add(r["Dispatch_Id"])
w = max(1.0, acc.get("SQ_WAVES", 0.0))
cyc = max(1.0, acc.get("SQ_WAVE_CYCLES", 0.0))
out = {"workload_key": sys.argv[3], "source_key": source_key(), "kernel_substr": sub, "launches": len(disp),
       "valu_per_wave": acc.get("SQ_INSTS_VALU", 0.0) / w, "salu_per_wave": acc.get("SQ_INSTS_SALU", 0.0) / w, "lds_per_wave": acc.get("SQ_INSTS_LDS", 0.0) / w,
       "wave_cycles_per_wave_quad": cyc / w, "wait_any_frac": acc.get("SQ_WAIT_ANY", 0.0) / cyc, "wait_inst_any_frac": acc.get("SQ_WAIT_INST_ANY", 0.0) / cyc,
       "active_inst_any_frac": acc.get("SQ_ACTIVE_INST_ANY", 0.0) / cyc, "waves_per_launch": w / max(1, len(disp)),
       "note": "SQ_* cycle counters are quad-cycles; one pass, program directly behind rocprofv3 --"}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
