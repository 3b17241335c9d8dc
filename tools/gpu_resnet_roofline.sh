#!/bin/bash
# Per-node roofline table of the ResNet3D graphs: tools/gpu_resnet_roofline.sh <tag>  ->  profiles/<tag>_resnet_node_roofline.txt
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rn && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rn -- python3 $R/tools/resnet_node_roofline.py run 12 > /tmp/rn.log 2>&1
tail -2 /tmp/rn.log
python3 $R/tools/resnet_node_roofline.py table /tmp/rn 12 > $R/profiles/${TAG}_resnet_node_roofline.txt
mkdir -p $R/gpurun_out/profiles_out && cp $R/profiles/${TAG}_resnet_node_roofline.txt $R/gpurun_out/profiles_out/
tail -40 $R/profiles/${TAG}_resnet_node_roofline.txt
