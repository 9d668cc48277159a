#!/bin/bash
# the scenes with computed textures, two builds alternating: tools/tex_scenes.sh "base cur"
libs="$1"
tools/abn.sh "$libs" 100 2 "--scene random_scene --width 1200 --height 800" || exit 1
tools/abn.sh "$libs" 256 2 "--scene two_spheres --width 800 --height 800" || exit 1
tools/abn.sh "$libs" 256 2 "--scene earth --width 800 --height 800" || exit 1
tools/abn.sh "$libs" 512 1 || exit 1
tools/abn.sh "$libs" 64 2 "--scene stress_volume --width 1920 --height 1080" || exit 1
