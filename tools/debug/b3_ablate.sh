# Build ablation variants of conv_b3.hip (compile-time switches) and time three layers with each: bash tools/debug/b3_ablate.sh
# (runs the builds HERE, the timings through gpurun)
set -e
cd /root/repo
mkdir -p variants
for V in BASE NOMFMA NOSPLIT NODMA "NOMFMA -DB3_ABL_NOSPLIT" "NOSPLIT -DB3_ABL_NODMA"; do
  N=$(echo $V | tr -d ' -' )
  if [ "$V" = BASE ]; then F=""; else F="-DB3_ABL_$V"; fi
  VITTA_EXTRA_CFLAGS="$F" python -m vitta_amd.build > /dev/null 2>&1
  cp vitta_amd/csrc/libvitta_hip.so variants/$N.so
done
python -m vitta_amd.build > /dev/null 2>&1
