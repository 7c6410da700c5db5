#!/bin/bash
# rocprofv3 counter passes over tools/pmc_kernels.py (run on the GPU box from the repo root):
#   bash tools/pmc_collect.sh gpurun_out/r2/pmc
# One pass per counter group (SQ has 8 slots, TCC 4; FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md
# "rocprofv3 PMC slots"), counters with --kernel-trace only, the program directly after `--`. Then:
#   python3 tools/pmc_summary.py gpurun_out/r2/pmc profiles/r02_pmc.json
set -euo pipefail
OUT=$(realpath -m "${1:-gpurun_out/pmc}")
ROOT=$(cd "$(dirname "$0")/.." && pwd)
# optional: another driver script of tools/ and its arguments (default: one ViT block, tools/pmc_kernels.py);
# PMC_PASSES limits the passes (default: all)
PROG=${2:-pmc_kernels.py}
shift $(( $# > 2 ? 2 : $# ))
PROG_ARGS=("$@")
PASSES=${PMC_PASSES:-"sq1 sq2 grbm tcc fetch write trace"}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() {  # name, counters...
  local name=$1; shift
  case " $PASSES " in *" $name "*) ;; *) return 0 ;; esac
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o "$name" -- python3 "$ROOT/tools/$PROG" "${PROG_ARGS[@]}" > "$OUT/$name.log" 2>&1
  echo "pass $name done"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_WAVES
run grbm GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
case " $PASSES " in *" trace "*)
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/tools/$PROG" "${PROG_ARGS[@]}" > "$OUT/trace.log" 2>&1
  echo "pass trace done" ;;
esac
