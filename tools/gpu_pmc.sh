# PMC passes at the batch sizes the bench runs (through gpurun from the repo root):  tools/gpu_pmc.sh TAG [FRAMES] [CONFIG] [SOLVER 1|0]
# three separate rocprofv3 --pmc passes (SQ counters; FETCH_SIZE; WRITE_SIZE -- MI355X_MICROARCH.md: never combined with
# other trace domains) of tools/pmc_target.py, summarised per kernel into gpurun_out/<TAG>_pmc_cfg<CONFIG>_<FRAMES>f.csv
TAG=${1:-r06}
CFG=${3:-2}
SOLVER=${4:-1}
SFX=$([ "$SOLVER" = 0 ] && echo _reference || echo "")
F=${2:-$([ "$CFG" = 5 ] && echo 128 || echo 1024)}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/prof_${TAG}_pmc_cfg${CFG}_${F}f${SFX}
rm -rf ${P}_*
for C in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d ${P}_$N -- python $R/tools/pmc_target.py $F $CFG $SOLVER 2>&1 | grep pmc_target
done
cd $R
python tools/pmc_summary.py ${P}_ > gpurun_out/${TAG}_pmc_cfg${CFG}_${F}f${SFX}.csv
cat gpurun_out/${TAG}_pmc_cfg${CFG}_${F}f${SFX}.csv | cut -c1-260
