R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_hiptrace
timeout 600 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $R/gpurun_out/prof_hiptrace -- python $R/tools/dev_pipeline_probe.py 6 2>&1 | tail -2
ls $R/gpurun_out/prof_hiptrace/*/ | head
