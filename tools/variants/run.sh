RANENV_LATE=2 RANENV_FUSE=3 timeout -k 10 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -2 | cut -c1-200 || exit 1
RANENV_SE_MODE=gather RANENV_FUSE=7 timeout -k 10 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -2 | cut -c1-200 || exit 1
RANENV_COMPACT=0 RANENV_FUSE=20 RANENV_ROW_WIDTH=16 timeout -k 10 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -2 | cut -c1-200 || exit 1
