cd /root/repo; export TMPDIR=/tmp
for spec in k35=build/k35/psxavenc_amd/libpsxav_hip.so new=cur; do
  name=${spec%%=*}; path=${spec#*=}; lib=""
  [ "$path" != cur ] && lib="$PWD/$path"
  for wl in "a4:--amp 4" "a8:--amp 8" "v3:--config sbs_v3 --total-frames 1250"; do
    w=${wl%%:*}; args=${wl#*:}
    out=gpurun_out/f2_s14_${name}_$w; rm -rf $out; mkdir -p $out
    PSXAV_HIP_LIB=$lib timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out -o r -- python bench.py --steps 2 --warmup 1 --launches-per-step 16 --lanes 1 --no-cpu-baseline --no-secondary $args > $out/log 2>&1
    python tools/rocpd_summary.py $(find $out -name '*.db') 2>/dev/null | grep "mdec_encode_frames" | grep "FETCH_SIZE" | awk -v n=$name -v w=$w '{printf "%s %s FETCH_SIZE %.1f MB (x2 = %.1f MB)\n", n, w, $NF/1024, $NF/512}'
  done
done
