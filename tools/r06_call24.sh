#!/bin/bash
# Round-6 call 24: LDS (bank conflicts) and L2 (hit rate) counter passes of the final fp32 engine
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out
rm -rf $O/pmc_lds $O/pmc_l2
PASSES="D E" bash tools/pmc_round.sh > $O/r06_ah_pmc_lds_l2.log 2>&1
python tools/pmc_summary.py $O/pmc_lds $O/pmc_lds.md > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_l2 $O/pmc_l2.md > /dev/null 2>&1
find $O/pmc_lds $O/pmc_l2 -type f -size +3M -delete
head -12 $O/pmc_lds.md | cut -c1-300; head -12 $O/pmc_l2.md | cut -c1-300
