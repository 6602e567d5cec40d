cd $GRAFT_REPO_ROOT
# the GPU suite with its files in reverse order (order dependence between files)
python -m pytest $(ls tests/test_*.py | tac) -m gpu -q -x 2>&1 | tail -5
