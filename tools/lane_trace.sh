cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r06/lane_trace
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r06/lane_trace -o t -- python3 tools/api_loop_host_split.py 1280 720 96 > gpurun_out/r06/lane_trace/run.log 2>&1
ls -la gpurun_out/r06/lane_trace/* | head; 
