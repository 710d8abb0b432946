import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"])
e=d.get("extra", d)
for k in ("msm","msm_k24"):
    if k in e:
        m=e[k]; print(k, "single", round(m["single_msm"]["ms_per_msm"],3), "batched", m.get("ms_per_msm_batched"), "| table:", {a:(round(b,3) if isinstance(b,float) else b) for a,b in m["over_shifted_base_table"].items() if a!="what"})
for k in ("create_proof","create_proof_k24"):
    if k in e: print(k, {kk:v for kk,v in e[k].items() if not isinstance(v,(dict,list,str))})
print("cpu", d.get("cpu_baseline"))
