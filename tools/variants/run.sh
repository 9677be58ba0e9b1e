for rep in 1 2; do
for m in stream gather; do
  export RANENV_SE_MODE=$m
  for f in 1 2 3 5 0; do
    echo "== $m FUSE=$f: $(RANENV_FUSE=$f timeout -k 10 100 python tools/kprobe.py 2>&1 | grep K= | tr '\n' ' ')"
  done
done
done
