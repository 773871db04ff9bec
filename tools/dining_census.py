import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import minimaloptix_amd as M
ctx = M.Context(0)
hs = M.HostScene("dining_standin", 1920, 1080, iarg=6); seeds = M.launch_seeds(16)
ctx.load(hs); ctx.accum_clear(); st = ctx.render_counted(seeds)
