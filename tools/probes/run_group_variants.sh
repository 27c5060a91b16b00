set -e
for v in default k2bg4 nogrmw; do
  if [ $v = default ]; then unset SASKTRAN2_AMD_LIB; else export SASKTRAN2_AMD_LIB=$PWD/sasktran2_amd/_lib/variants/$v.so; fi
  echo "==== $v"
  timeout -k 10 300 python tools/probes/c2_group_cost.py 10000 1 2>&1 | grep -E "k1g|k2b_adj|k2c_adj|solve" | sed "s/.*upload [0-9.]* ms; //; s/radiance out.*//"
done
