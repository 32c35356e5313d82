#!/bin/bash
# k_train_dgrad3 / k_train_app3 with phases switched off (results wrong by construction; only the times mean something):
# TRAIN_ENG bit 32 = no row stores, 64 = no position gradient / X (app3), 128 = no dz1 products (dgrad3)
export TMPDIR=/tmp
for e in 1 33 65 97 129; do
  TRAIN_ENG=$e bash scripts/serial_trace.sh dg$e > /dev/null 2>&1
  echo "TRAIN_ENG=$e: $(grep -E 'k_train_dgrad3' gpurun_out/serial_dg$e.md | cut -d'|' -f3-7) || $(grep -E 'k_train_app3' gpurun_out/serial_dg$e.md | cut -d'|' -f3-7) $(grep 'fwd+bwd' gpurun_out/serial_dg$e.log)"
done
