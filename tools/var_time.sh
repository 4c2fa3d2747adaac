cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in tools/var/libvar4.so; do
  rm -rf gpurun_out/vt; FPC_VARIANT_LIB=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vt -- python3 tools/nn_only.py 4 > gpurun_out/vt.log 2>&1
  echo "variant [$v]"; grep -E "k_tower|k_fc256" gpurun_out/vt/*/*kernel_stats.csv | cut -d, -f1-4
done
