export TMPDIR=/tmp
for o in 0 1 3 5 7 9; do echo -n "OVL2=$o: "; OVL2=$o python -u scripts/train_serial_probe.py 2>&1 | grep "fwd+bwd"; done
