#!/bin/bash
# Round 5, batch P (closing): full GPU suite on the final build, then the round's profile set without the PMC passes (cell
# kernels unchanged since the committed ones)
export SKIP_PMC=1
bash tools/r05/batch_d.sh
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
