cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/fz4; mkdir -p $o
(echo "# third campaign, final tree (sources $(python tests/tools/tree_hash.py)): FUZZ_BIG=1 fuzz.py 330 9202; fuzz.py 200 9201"
 FUZZ_BIG=1 timeout 500 python tests/tools/fuzz.py 330 9202 2>&1 | tail -1
 timeout 400 python tests/tools/fuzz.py 200 9201 2>&1 | tail -1) > $o/fuzz4.txt 2>&1
