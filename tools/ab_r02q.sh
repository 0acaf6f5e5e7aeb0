# copy stream priority: does the e2e rate without a tracker depend on which hardware queue the upload stream lands in?
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for pr in 0 -1; do
  echo "copy priority $pr: $(SLEAP_AMD_COPY_STREAM_PRIORITY=$pr timeout 300 python tools/predict_e2e.py 2560 arrays 2>&1 | grep 'frames/s' | tail -2 | sed 's/.*= //' | tr '\n' ' ')"
done; done
