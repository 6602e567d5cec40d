cd $GRAFT_REPO_ROOT
examples/_build/custom_potential | tail -3
python -m pytest tests/test_cxx_interface.py -m gpu -q -x -k "parameter_updatable or examples_run" 2>&1 | tail -3
