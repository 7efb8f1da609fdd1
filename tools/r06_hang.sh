#!/bin/bash
# round 6: a random-parity problem that stopped the stress run (seed 1672): again, several times, with a deadline
for i in 1 2 3 4 5 6; do
  PYTHONFAULTHANDLER=1 timeout -s ABRT -k 5 60 python tests/stress/random_parity.py ${1:-1672} ${2:-1673} > gpurun_out/hang_$i.txt 2>&1
  echo "run $i rc $? $(grep -v '^  File\|^$' gpurun_out/hang_$i.txt | tail -3 | cut -c1-150 | tr '\n' ' ')"
done
