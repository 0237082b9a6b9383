"""pmc_*.json of `tools/probe.sh requests` -> requests per kernel and per chain: fabric reads / writes / atomics
(TCC_EA0_RDREQ / WRREQ / ATOMIC), L2 requests / hits.  One profiled step = one chain per kernel launch set; the figure
per chain is (avg per launch) x (launches of that kernel in one chain), taken from the launch counts themselves."""
import json
import sys

merged = {}
for f in sys.argv[1:]:
    for k, c in json.load(open(f)).items():
        merged.setdefault(k, {}).update(c)
own = {k: v for k, v in merged.items() if not k.startswith(("__amd_rocclr", "at::", "copy_kernel", "burn_"))}
chains = max(1, min(v["TCC_EA0_RDREQ_sum"]["launches"] for k, v in own.items() if k.startswith("frame_init_kernel")))
rows = {}
tot = {"fabric_rd": 0.0, "fabric_wr": 0.0, "fabric_atomic": 0.0, "l2_req": 0.0, "l2_hit": 0.0}
for k, v in sorted(own.items()):
    g = lambda n: v.get(n, {}).get("avg", 0.0) * v.get(n, {}).get("launches", 0) / chains
    r = {"launches_per_chain": v["TCC_EA0_RDREQ_sum"]["launches"] / chains, "fabric_rd": g("TCC_EA0_RDREQ_sum"),
         "fabric_wr": g("TCC_EA0_WRREQ_sum"), "fabric_atomic": g("TCC_EA0_ATOMIC_sum"), "l2_req": g("TCC_REQ_sum"),
         "l2_hit": g("TCC_HIT_sum")}
    rows[k] = {a: round(b, 1) for a, b in r.items()}
    for a in tot:
        tot[a] += r[a]
out = {"what": "memory requests per launch chain (rocprofv3 --pmc, one context, one chain per step), own kernels only",
       "chains_profiled": chains, "per_chain_total": {a: round(b) for a, b in tot.items()},
       "fabric_requests_per_chain": round(tot["fabric_rd"] + tot["fabric_wr"]), "kernels": rows}
print(json.dumps(out, indent=1))
print(json.dumps({"fabric_requests_per_chain_M": round((tot["fabric_rd"] + tot["fabric_wr"]) / 1e6, 2),
                  "atomics_M": round(tot["fabric_atomic"] / 1e6, 2), "l2_req_M": round(tot["l2_req"] / 1e6, 2)}))
