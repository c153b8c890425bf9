# per-launch roofline table of one single-stream forward: bash tools/op_times.sh [dtype=fp16] -> gpurun_out/op_times_<dtype>.txt
DT=${1:-fp16}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/_ops && mkdir -p gpurun_out/_ops
VTI_SINGLE_STREAM=1 VTI_LIST_OPS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_ops -- python3 tools/prof_forward.py 64 $DT 5 2> gpurun_out/_ops/ops.txt > /dev/null && python3 tools/op_times.py gpurun_out/_ops 64 $DT > gpurun_out/op_times_$DT.txt
