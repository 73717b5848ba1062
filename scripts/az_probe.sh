#!/bin/bash
# self-play: full JSON lines of the product build and cycle stamps of the instrumented one for "E S [variant]" triples.  GPU box, repo root.
out=$PWD/gpurun_out/${AZ_OUT:-azp}; mkdir -p $out
for cfg in "$@"; do set -- $cfg
  echo "== $1 x $2 variant ${3:-0}" | tee -a $out/log.txt
  python3 scripts/bench_az.py --envs $1 --searches $2 --variant ${3:-0} 2>/dev/null | tee -a $out/log.txt
  TW_ABLATE=1 TW_STAMPS=1 python3 scripts/bench_az.py --envs $1 --searches $2 --steps 1 --variant ${3:-0} 2>&1 | grep -v amdgpu.ids | grep -v '^{' | tail -4 | tee -a $out/log.txt
done
