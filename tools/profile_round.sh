#!/bin/bash
# Run on the GPU box (via gpurun): the round's evidence for one state of the code.
#   tools/profile_round.sh TAG [plain]   (plain: bench.py --plain, the workload of rounds 1-2; default: the 8(d) mix)  ->  gpurun_out/TAG_bench10M.json, TAG_kernel_stats_bench10M.txt,
#                                    TAG_pmc_fetch_write_bench10M.txt, TAG_pmc_traffic.json
# Counter passes are separate from each other and carry only --kernel-trace (see MI355X_MICROARCH.md).
set -u
tag=$1
extra=""; cfgx=', "multihit_frac": 0.05, "indel_frac": 0.03, "max_copies": 41'
if [ "${2:-}" = plain ]; then extra="--plain"; cfgx=''; fi
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
timeout 1500 python bench.py $extra --detail $out/${tag}_bench_detail.json > $out/${tag}_bench10M.json 2> $out/${tag}_bench10M.err
rm -rf /tmp/prof_ks /tmp/prof_f /tmp/prof_w
(cd /tmp && THJ_BENCH_NO_REPLAY=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o res -- python $root/bench.py $extra --steps 5 --warmup 1 --no-cpu-baseline --e2e-pairs 0 --no-pmc --detail /tmp/bench_detail_prof.json > /tmp/prof_ks.log 2>&1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py $extra --steps 5 --warmup 1 --no-cpu-baseline --e2e-pairs 0   (MI355X, $tag)";
  echo "# durations in microseconds"; python tools/rocpd_summary.py $(find /tmp/prof_ks -name '*.db' | head -1); } > $out/${tag}_kernel_stats_bench10M.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -o res -- python $root/bench.py $extra --steps 2 --warmup 1 --no-cpu-baseline --e2e-pairs 0 --no-pmc --detail /tmp/bench_detail_prof.json > /tmp/prof_f.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -o res -- python $root/bench.py $extra --steps 2 --warmup 1 --no-cpu-baseline --e2e-pairs 0 --no-pmc --detail /tmp/bench_detail_prof.json > /tmp/prof_w.log 2>&1)
{ echo "# rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline  (separate pass; KB per dispatch)";
  python tools/rocpd_summary.py $(find /tmp/prof_f -name '*.db' | head -1) thj_k;
  echo; echo "# rocprofv3 --pmc WRITE_SIZE --kernel-trace -- (same command, separate pass)";
  python tools/rocpd_summary.py $(find /tmp/prof_w -name '*.db' | head -1) thj_k; } > $out/${tag}_pmc_fetch_write_bench10M.txt
python - "$out/${tag}_pmc_fetch_write_bench10M.txt" "$tag" "$cfgx" > $out/${tag}_pmc_traffic.json <<'PY'
import json, re, sys
ker = {}
for line in open(sys.argv[1]):
    m = re.match(r"(?:void )?(thj_k_\w+)(<[^>]*>)?\(.*?\s+(FETCH_SIZE|WRITE_SIZE)\s+\d+\s+[\d.]+\s+([\d.]+)\s*$", line)
    if m:       # under the name with its template arguments, and -- the instances together: each is launched once per side -- without
        for key in {m.group(1), m.group(1) + (m.group(2) or "")}:
            d = ker.setdefault(key, {})
            d[m.group(3)] = d.get(m.group(3), 0.0) + float(m.group(4))
cfg = json.loads('{"pairs_per_gpu": 10000000, "genome_len": 64444167, "exon_len": 300' + sys.argv[3] + '}')
print(json.dumps({"config": cfg, "unit": "KB per dispatch",
                  "source": "profiles/%s_pmc_fetch_write_bench10M.txt" % sys.argv[2], "kernels": ker}, indent=1))
PY
tail -1 $out/${tag}_bench10M.json | cut -c1-300
