t() { echo "$1: $(env $1 python tools/time_scenes.py book2 800 800 100 2>&1 | grep -v "^\[rtg\]" | cut -c28-48)"; }
t X=0
for v in 12 16 24 28 32; do t RTG_REFILL_MIN=$v; done
for v in 32 36 44 48 56; do t RTG_GATHER_MIN=$v; done
for v in 8 16 32 40; do t RTG_SPHERE_MIN=$v; done
for v in 16 24 48 64; do t RTG_BOX_LEAVE=$v; done
for v in 4 8 12 24 32; do t RTG_RUN_AHEAD=$v; done
for v in 8 16 32 40; do t RTG_RUN_AHEAD_MIN=$v; done
t X=0
