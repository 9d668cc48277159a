#!/bin/bash
# one line per configuration of the baseline table (one MI355X)
timeout -k 10 100 python tools/render_once.py --scene random_scene --width 1200 --height 800 --spp 100 --passes 2 || exit 1
timeout -k 10 100 python tools/render_once.py --scene cornell_box --width 600 --height 600 --spp 256 --passes 2 || exit 1
timeout -k 10 100 python tools/render_once.py --scene final_scene --width 800 --height 800 --spp 256 --passes 2 || exit 1
timeout -k 10 200 python tools/render_once.py --scene spheres_grid --n 1000000 --width 3840 --height 2160 --spp 16 --passes 2 || exit 1
timeout -k 10 100 python tools/render_once.py --scene stress_volume --width 1920 --height 1080 --spp 64 --passes 2 || exit 1
timeout -k 10 100 python tools/render_once.py --scene my_scene --width 800 --height 800 --spp 32 --passes 2 || exit 1
