#!/bin/bash
# Everything the round-end driver runs, in its order (developer convenience).  On a box without a GPU the GPU steps are skipped.
set -e -o pipefail
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build(); print('build ok')"
python -m pytest tests -x -q -m "not gpu"
if python -c "import torch, sys; sys.exit(0 if torch.cuda.is_available() else 1)" 2>/dev/null; then
  python -m pytest tests -x -q -m gpu
  python -c "import __graft_entry__ as g; g.smoke()"
  python bench.py
else
  echo "no GPU here: skipped pytest -m gpu, smoke() and bench.py (use gpurun)"
fi
