cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace8 -- python3 $GRAFT_REPO_ROOT/tools/rehearse_rank.py --world 8 --epochs 12 --warmup 4 > $GRAFT_REPO_ROOT/gpurun_out/trace8.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/trace8 | head
