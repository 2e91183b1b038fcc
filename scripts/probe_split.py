import sys, torch; sys.path.insert(0, ".")
from neighbour_feature_pooling_amd import NFPPooling
from bench import time_kernel_graph
s = torch.cuda.Stream()
m = NFPPooling(512, R=1, measure="cosine", padding=1)
for shape in [(64,512,7,7),(128,512,5,7),(128,512,4,7),(256,512,4,7),(256,512,3,7),(192,512,3,7)]:
    x = torch.randn(*shape, device="cuda")
    with torch.cuda.stream(s), torch.no_grad():
        t = time_kernel_graph(lambda: m(x), 50, s)
    print(shape, f"fwd {t:.2f} us")
