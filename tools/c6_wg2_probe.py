import sys; sys.argv=['x','--time-only']
sys.path[:0]=['/root/repo/tools']
import importlib.util, os
spec=importlib.util.spec_from_file_location('c6', 'tools/conv6_check.py'); c6=importlib.util.module_from_spec(spec)
sys.modules['c6']=c6
# do not run main
src=open('tools/conv6_check.py').read().split('if __name__ == "__main__":')[0]
exec(compile(src,'c6','exec'), c6.__dict__)
import hdmoe_hip; hdmoe_hip.lib()
for (N,R,Ci,Co) in [(512,32,32,32),(512,32,64,32),(512,32,96,32)]:
    c6.timeit(N,R,Ci,Co,[3,3,5,5])
c6.timeit(512,32,32,32,[3]); c6.timeit(512,32,32,32,[5])
