# round 6, GPU call 3: the whole GPU tier as the driver runs it (+ -rs through pytest.ini), then smoke()
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6c; mkdir -p $O; export LCC_PARITY_OUT=$GRAFT_REPO_ROOT/$O
( time python -m pytest tests/ -x -q -m gpu 2>&1 | tail -40 ) > $O/pytest_gpu_tail.txt 2>&1
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/smoke_tail.txt 2>&1
tail -45 $O/pytest_gpu_tail.txt; cat $O/smoke_tail.txt
