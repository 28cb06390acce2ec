# rocprofv3 kernel stats of tools/pmc_target.py (synchronous batches, one alone on the chip): tools/gpu_kstats.sh TAG [FRAMES] [CONFIG]
TAG=${1:-r06}; CFG=${3:-2}; SOLVER=${4:-1}; SFX=$([ "$SOLVER" = 0 ] && echo _reference || echo ""); F=${2:-$([ "$CFG" = 5 ] && echo 64 || echo 1024)}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/prof_${TAG}_kstats_cfg${CFG}_${F}f${SFX}
rm -rf $D
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python $R/tools/pmc_target.py $F $CFG $SOLVER 2>&1 | grep pmc_target
cd $R
find $D -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "cp {} gpurun_out/${TAG}_kstats_cfg${CFG}_${F}f${SFX}.csv; cut -c1-150 {} | head -24"
