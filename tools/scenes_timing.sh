#!/bin/bash
# timing of the scenes used to judge a change, one line each
timeout -k 10 100 python tools/render_once.py --spp 64 --passes 2 || exit 1
timeout -k 10 100 python tools/render_once.py --scene stress_volume --spp 32 --passes 2 || exit 1
timeout -k 10 100 python tools/render_once.py --scene random_scene --spp 32 --passes 2 || exit 1
timeout -k 10 100 python tools/render_once.py --scene cornell_box --spp 64 --passes 2 || exit 1
timeout -k 10 200 python tools/render_once.py --scene spheres_grid --n 1000000 --spp 4 --passes 2 || exit 1
timeout -k 10 200 python tools/render_once.py --scene my_scene --spp 32 --passes 2 || exit 1
