# PMC passes over the GEMM probe: bash tools/probe_pmc.sh <tag> [nt|tn]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
KIND=${2:-nt}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PROBE_M=${PROBE_M:-98304}
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1 -- python3 $R/tools/gemm_probe.py $KIND 3 > $OUT/p1.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY --output-format csv -d $OUT/p2 -- python3 $R/tools/gemm_probe.py $KIND 3 > $OUT/p2.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/p3 -- python3 $R/tools/gemm_probe.py $KIND 3 > $OUT/p3.log 2>&1 || exit 1
find $OUT -name "*counter_collection.csv"
