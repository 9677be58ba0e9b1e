timeout -k 10 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 | cut -c1-200 || exit 1
timeout -k 10 500 python tools/abprobe.py tools/variants/cur.so@FUSE=1 tools/variants/cur.so tools/variants/d2.so@FUSE=1 tools/variants/d2.so 2>&1 | tail -4
for l in cur d2; do echo "== $l kprobe: $(RANENV_LIB=$PWD/tools/variants/$l.so timeout -k 10 100 python tools/kprobe.py 2>&1 | grep K= | tr '\n' ' ')"; done
for l in cur d2; do echo "== $l pipeprobe: $(RANENV_LIB=$PWD/tools/variants/$l.so timeout -k 10 100 python tools/pipeprobe.py 2 300 stream 2>&1 | grep "us per TTI" | tr '\n' ' ')"; done
