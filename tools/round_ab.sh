#!/bin/bash
# Every named scene at its benchmark size and pass size, two builds alternating on one box: tools/round_ab.sh "base cur"
libs="$1"
tools/abn.sh "$libs" 1536 1 || exit 1
tools/abn.sh "$libs" 768 1 "--scene random_scene --width 1200 --height 800" || exit 1
tools/abn.sh "$libs" 1536 1 "--scene cornell_box --width 600 --height 600" || exit 1
tools/abn.sh "$libs" 64 1 "--scene spheres_grid --n 1000000 --width 3840 --height 2160" || exit 1
tools/abn.sh "$libs" 384 1 "--scene stress_volume --width 1920 --height 1080" || exit 1
tools/abn.sh "$libs" 1536 1 "--scene my_scene --width 800 --height 800" || exit 1
tools/abn.sh "$libs" 1536 1 "--scene cornell_smoke --width 600 --height 600" || exit 1
for sc in two_spheres two_perlin_spheres earth simple_light; do tools/abn.sh "$libs" 1024 1 "--scene $sc --width 800 --height 800" || exit 1; done
