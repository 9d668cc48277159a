#!/bin/bash
libs="$1"
tools/abn.sh "$libs" 512 2 || exit 1
tools/abn.sh "$libs" 256 2 "--scene cornell_box --width 600 --height 600" || exit 1
tools/abn.sh "$libs" 64 2 "--scene stress_volume --width 1920 --height 1080" || exit 1
tools/abn.sh "$libs" 32 2 "--scene my_scene --width 800 --height 800" || exit 1
tools/abn.sh "$libs" 100 2 "--scene random_scene --width 1200 --height 800" || exit 1
tools/abn.sh "$libs" 16 2 "--scene spheres_grid --n 1000000 --width 3840 --height 2160" || exit 1
