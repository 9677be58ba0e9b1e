timeout -k 10 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -2 | cut -c1-200 || exit 1
for m in stream gather; do export RANENV_SE_MODE=$m; timeout -k 10 500 python tools/abprobe.py tools/variants/prev.so tools/variants/div.so 2>&1 | tail -2; done
