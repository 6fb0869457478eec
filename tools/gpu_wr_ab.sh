cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for q in 0 1; do
  if [ $q = 1 ]; then export PSXHIP_MDEC_NO_RETRY_QUEUE=1; else unset PSXHIP_MDEC_NO_RETRY_QUEUE; fi
  out=gpurun_out/wr_ab_$q; rm -rf $out; mkdir -p $out
  PSXHIP_MDEC_STATS=0 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VMEM_WR --kernel-trace -d $out/w -o r -- python tools/gpu_mdec_probe.py a4 > $out/log 2>&1
  python tools/rocpd_summary.py $(find $out -name '*.db') 2>/dev/null | grep "mdec_encode" | grep -v "^void.*calls"
done
