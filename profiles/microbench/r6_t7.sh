cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
ulimit -c 0
bash profiles/mfma_util.sh r06_deepfm_wide --hidden 1024,512,256
bash profiles/mfma_util.sh r06_mmoe --model mmoe
bash profiles/mfma_util.sh r06_xdeepfm --model xdeepfm
bash profiles/r06_lines.sh
