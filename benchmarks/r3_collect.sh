# copies the results of benchmarks/r3_profiles.sh (gpurun_out/r3p) into profiles/ and prints a summary
set -e
O=gpurun_out/r3p
for k in c1 c2 c3 c5 c4_iters50 c4_iters100 c4_iters300; do cp $O/bench_$k.json profiles/r3_bench_$k.json; done
python3 benchmarks/trim_stats.py $O/prof_c2 profiles/r3_bench_c2_stream_kernel_stats.csv
python3 benchmarks/trim_stats.py $O/prof_c2f profiles/r3_bench_c2_fused_kernel_stats.csv
python3 benchmarks/trim_stats.py $O/prof_c3 profiles/r3_bench_c3_stream_kernel_stats.csv
python3 benchmarks/trim_stats.py $O/prof_c3f profiles/r3_bench_c3_fused_kernel_stats.csv
python3 benchmarks/trim_stats.py $O/prof_c4 profiles/r3_bench_c4_iters50_kernel_stats.csv
python3 benchmarks/trim_stats.py $O/prof_c5 profiles/r3_bench_c5_kernel_stats.csv
for k in c2 c2f c3 c3f c4 c5; do cp $O/prof_$k.json profiles/r3_profiled_bench_$k.json; done
cp $O/pmc_mfma_c2f.csv profiles/r3_pmc_mfma_c2_fused.csv
GVAE_PMC_ROUND=r3 python3 benchmarks/pmc_traffic.py c2 $O/pmc_c2_FETCH_SIZE $O/pmc_c2_WRITE_SIZE mpm_stream_kernel 32 3743 "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 benchmarks/matching_bench.py 32 256 50 1 1" > /dev/null
GVAE_PMC_ROUND=r3 python3 benchmarks/pmc_traffic.py c3 $O/pmc_c3_FETCH_SIZE $O/pmc_c3_WRITE_SIZE mpm_stream_kernel 0 1230 "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 benchmarks/matching_ragged.py" 17672135008 > /dev/null
GVAE_PMC_ROUND=r3 python3 benchmarks/pmc_traffic.py c5 $O/pmc_c5_FETCH_SIZE $O/pmc_c5_WRITE_SIZE mpm_stream_kernel 128 800 "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 benchmarks/matching_bench.py 128 16 50 0 1" > /dev/null
python3 - <<'PY'
import json, glob, csv
for f in sorted(glob.glob("gpurun_out/r3p/bench_*.json")):
    d = json.load(open(f)); r = d["roofline"]; fu = d.get("fused") or {}; fr = fu.get("roofline") or {}
    ms = r.get("matching_step") or {}
    print(f.split("/")[-1], "value %.1f ms %.3f frac %.3f launch %.3f ms it %.1f/%.1f tail %.3f es %.0f GB/s mstep %.3f | fused %s %s | fk %s ms %s TF | cpu %s %s" % (
        d["value"], d["ms_per_step"], r["frac"], r["avg_launch_ms"], r["streamed_iters_per_graph"], r["mean_mpm_iters_per_graph"],
        r["tail_sfree_ms_per_step"], r.get("edge_sim_GBps", 0), ms.get("frac", 0), fu.get("value"), fu.get("ms_per_step"), fr.get("avg_launch_ms"), fr.get("achieved"),
        (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline_1thread") or {}).get("value")))
t = json.load(open("profiles/pmc_traffic.json"))
print({k: round(v["ratio_to_algorithmic"], 4) for k, v in t.items()})
for f in sorted(glob.glob("profiles/r3_bench_*_kernel_stats.csv")):
    print(f)
    for r in list(csv.DictReader(open(f)))[:40]:
        if "gvae" in r["Name"]:
            n = int(r["Calls"]); tt = float(r["TotalDurationNs"])
            print(f"  {r['Name'].split('(')[0][-44:]:44s} {n:5d} calls {tt / n / 1e3:9.1f} us")
print(open("profiles/r3_pmc_mfma_c2_fused.csv").read()[:1800])
PY
