cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ser
for sh in 0 1 2; do
  export CBLX_SERDE_SHAPE=$sh
  echo "== shape $sh"
  rocprofv3 --kernel-trace --stats -d gpurun_out/ser/prof$sh -o s -- python3 tools/dev_serialize_rate.py > gpurun_out/ser/rate$sh.log 2>&1
  grep "same buffer\|pinned" gpurun_out/ser/rate$sh.log | tail -2
  python3 tools/rocpd_summary.py gpurun_out/ser/prof$sh/s_results.db | grep -i "serde_bucket<.*true, false>" | head -4
done
