cd $GRAFT_REPO_ROOT
time examples/_build/lj_benchmark 131072 30000 64 2>&1 | tail -4
cd /tmp; rm -rf ljlong; mkdir ljlong; cd ljlong
timeout 60 $GRAFT_REPO_ROOT/examples/_build/ref_LJMultipleTypes 2>&1 | tail -2
timeout 100 $GRAFT_REPO_ROOT/examples/_build/ref_BDHI > out.txt 2> err.txt; echo "BDHI rc=$? lines=$(wc -l < out.txt) $(grep -ci 'nan' out.txt) nan; $(tail -1 err.txt | cut -c1-150)"
timeout 100 $GRAFT_REPO_ROOT/examples/_build/ref_customPotentials > out2.txt 2> err2.txt; echo "customPotentials rc=$? $(tail -2 err2.txt | cut -c1-200)"; ls
