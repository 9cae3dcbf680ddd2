#!/bin/bash
# Profile collection on the GPU box (run through gpurun): `ROUND=r04 tools/profile_round.sh stats|pmc|pcnpmc|wide [tag]`.  Kernel traces and
# PMC counters are collected in SEPARATE rocprofv3 runs (never --pmc together with a trace domain other than --kernel-trace/--stats);
# the program follows `--` directly.  Raw traces are summarised on the box and deleted (gpurun copies back at most 64 MiB).
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
MODE=${1:-stats}
O=gpurun_out/${ROUND:-r04}_prof_${2:-a}
mkdir -p $O
summ_stats() {  # dir tag
  local db=$(find $1 -name "*.db" | head -1)
  if [ -n "$db" ]; then python3 tools/rocpd_stats.py "$db" $O/$2_kernel_stats.csv; else cp $(find $1 -name "*kernel_stats.csv" | head -1) $O/$2_kernel_stats.csv; fi
  rm -rf $1
}
summ_pmc() {   # dir tag
  local db=$(find $1 -name "*.db" | head -1)
  local csv=$(find $1 -name "*counter_collection.csv" | head -1)
  if [ -n "$csv" ]; then python3 tools/pmc_summary.py "$csv" > $O/$2.txt; else python3 tools/pmc_summary_db.py "$db" > $O/$2.txt; fi
  rm -rf $1
}
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --strong-images 0"
if [ "$MODE" = "stats" ] || [ "$MODE" = "all" ]; then
echo "== kernel stats of the bench command $(date +%T)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats_bench -o bench -- $B --no-variants --throughput-images 0 > $O/stats_bench.log 2>&1 || exit 1
summ_stats $O/stats_bench bench
echo "== kernel stats of the PCN fits $(date +%T)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_pcn -o pcn -- python3 tools/kbench_pcn.py --steps 200 > $O/stats_pcn.log 2>&1 || exit 1
summ_stats $O/stats_pcn pcn
echo "== kernel stats of the CDN fit $(date +%T)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_cdn -o cdn -- python3 tools/kbench_cdn.py > $O/stats_cdn.log 2>&1 || exit 1
summ_stats $O/stats_cdn cdn
echo "== kernel stats of the joint step $(date +%T)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_joint -o joint -- python3 tools/kbench_joint.py > $O/stats_joint.log 2>&1 || exit 1
summ_stats $O/stats_joint joint
fi
if [ "$MODE" = "flowstats" ]; then   # the two composite fits only
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_pcn -o pcn -- python3 tools/kbench_pcn.py --steps 200 > $O/stats_pcn.log 2>&1 || exit 1
summ_stats $O/stats_pcn pcn
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_cdn -o cdn -- python3 tools/kbench_cdn.py > $O/stats_cdn.log 2>&1 || exit 1
summ_stats $O/stats_cdn cdn
fi
if [ "$MODE" = "pmc" ] || [ "$MODE" = "all" ]; then
S="python3 bench.py --steps 1 --warmup 0 --epochs 50 --kernel-iters 20 --no-cpu-baseline --throughput-images 0 --no-variants --strong-images 0"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c (bench) $(date +%T)"; timeout -k 10 300 rocprofv3 --pmc $c -d $O/pmc_$c -o p -- $S > $O/pmc_$c.log 2>&1 || exit 1
  summ_pmc $O/pmc_$c bench_pmc_$c
done
echo "== pmc SQ set 1 (bench) $(date +%T)"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/pmc_sq1 -o p -- $S > $O/pmc_sq1.log 2>&1 || exit 1
summ_pmc $O/pmc_sq1 bench_pmc_sq
fi
if [ "$MODE" = "pcnpmc" ] || [ "$MODE" = "all" ]; then
P="python3 tools/kbench_pcn.py --steps 30 --case xyt"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c (pcn xyt) $(date +%T)"; timeout -k 10 300 rocprofv3 --pmc $c -d $O/pcn_$c -o p -- $P > $O/pcn_$c.log 2>&1 || exit 1
  summ_pmc $O/pcn_$c pcn_pmc_$c
done
echo "== pmc SQ (pcn xyt) $(date +%T)"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/pcn_sq -o p -- $P > $O/pcn_sq.log 2>&1 || exit 1
summ_pmc $O/pcn_sq pcn_pmc_sq
fi
if [ "$MODE" = "cdnpmc" ]; then   # coupling-flow point kernels at (1,1) and at the shipped split shapes: kernel stats + SQ counters
for shp in 11 0; do
  if [ "$shp" = "0" ]; then unset INR_FLOW_SHAPE; else export INR_FLOW_SHAPE=$shp; fi
  echo "== cdn stats shape $shp $(date +%T)"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/cdn_stats_$shp -o cdn -- python3 tools/kbench_cdn.py > $O/cdn_stats_$shp.log 2>&1 || exit 1
  summ_stats $O/cdn_stats_$shp cdn_shape$shp
  echo "== cdn pmc SQ shape $shp $(date +%T)"
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/cdn_sq_$shp -o p -- python3 tools/kbench_cdn.py --steps 30 > $O/cdn_sq_$shp.log 2>&1 || exit 1
  summ_pmc $O/cdn_sq_$shp cdn_pmc_sq_shape$shp
done
fi
if [ "$MODE" = "wide" ] || [ "$MODE" = "all" ]; then
echo "== kernel stats of the layer-by-layer path $(date +%T)"
for shp in ${WIDE_SHAPES:-256x1}; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_wide -o wide -- python3 tools/kbench_wide.py $shp > $O/stats_wide_$shp.log 2>&1 || exit 1
summ_stats $O/stats_wide wide_h${shp/x/_L}
done
fi
rm -f $O/*.log.tmp; du -sh $O; ls -la $O
