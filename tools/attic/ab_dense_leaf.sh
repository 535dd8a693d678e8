mkdir -p gpurun_out/r05_leaf
(
for sc in clutter plain; do
    echo "== new $sc"; RF_SCENE_DETAIL=$sc python tools/gpu_dense_leaf.py 32 dense_leaf_min=0 dense_leaf_min=3,leaf_vote=20 dense_leaf_min=3,leaf_vote=14 dense_leaf_min=3,leaf_vote=26 dense_leaf_min=3,leaf_vote=32 dense_leaf_min=2,leaf_vote=20 dense_leaf_min=5,leaf_vote=20 2>&1 | grep -v amdgpu.ids | tail -8
done
) > gpurun_out/r05_leaf/fourth.log 2>&1
cat gpurun_out/r05_leaf/fourth.log
