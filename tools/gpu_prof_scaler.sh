cd /root/repo; export TMPDIR=/tmp
out=gpurun_out/prof_scaler2; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $out/sq -o r -- python tools/gpu_frontend_bench.py > $out/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace -d $out/sq2 -o r -- python tools/gpu_frontend_bench.py > $out/sq2.log 2>&1
python tools/rocpd_summary.py $(find $out -name '*.db' | sort) | grep -E "scaler_kernel" | cut -c1-120
