#!/bin/bash
# base (old library AND old compiler: the scene is compiled by the library in use) against cur on the small scenes
for sc in "cornell_smoke --width 600 --height 600" "simple_light --width 800 --height 800" "two_spheres --width 800 --height 800" "earth --width 800 --height 800" "two_perlin_spheres --width 800 --height 800"; do
  echo "== $sc"; tools/abn.sh "base cur" 256 2 "--scene $sc" || exit 1
done
