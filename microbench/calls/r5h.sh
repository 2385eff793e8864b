O=gpurun_out/r5h; mkdir -p $O
for i in 1 2 3; do
  JPEGX_LIB_PATH=microbench/_ab/libjpegx_oldsz.so python bench.py --no-cpu-baseline > $O/old$i.json 2>> $O/err.txt || exit 1
  python bench.py --no-cpu-baseline > $O/new$i.json 2>> $O/err.txt || exit 1
done
