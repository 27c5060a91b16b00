for cfg in "SASKTRAN2_AMD_HOST_ALLOC=pageable" "SASKTRAN2_AMD_HOST_FLAGS=0x80000000" "SASKTRAN2_AMD_HOST_FLAGS=0x40000000" "SASKTRAN2_AMD_HOST_FLAGS=0x3"; do
  echo "== $cfg"
  env $cfg python tools/probes/host_path_phases.py 2>&1 | grep -E "host phases|whole call" | tail -4
done
