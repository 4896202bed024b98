"""Which fork/join shapes does hipGraph capture (via torch.cuda.graph) accept on this box?"""
import sys, faulthandler
faulthandler.enable()
import torch
Stream = torch.cuda.Stream
case = sys.argv[1]
a = torch.zeros(1 << 20, device="cuda")
s1, s2, s3 = Stream(), Stream(), Stream()
warm = Stream()
warm.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(warm):
    a.add_(1)
torch.cuda.current_stream().wait_stream(warm)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    if case == "flat":            # fork two, join
        for s in (s1, s2):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                a.add_(1)
        for s in (s1, s2):
            cur.wait_stream(s)
    elif case == "nested":        # fork s1 from cur, fork s2 from s1
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            a.add_(1)
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                b = a * 2
            s1.wait_stream(s2)
            a.add_(1)
        cur.wait_stream(s1)
    elif case == "refork":        # same side stream forked twice from cur with work in between
        for _ in range(3):
            s1.wait_stream(cur)
            with torch.cuda.stream(s1):
                a.add_(1)
            cur.wait_stream(s1)
            a.add_(1)
    elif case == "late_join":     # aux forked from cur, cur continues, join later
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            b = a * 2
        a2 = a + 1
        a3 = a2 + 1
        cur.wait_stream(s1)
    elif case == "aux_refork_nojoin_between":   # aux forked twice from cur before one join
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            b = a * 2
        a2 = a + 1
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            c = a2 * 2
        cur.wait_stream(s1)
    elif case == "prefork":       # both side streams enter the capture from cur first; then s2 depends on s1
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            a.add_(1)
        s2.wait_stream(s1)
        with torch.cuda.stream(s2):
            b = a * 2
        s1.wait_stream(s2)
        with torch.cuda.stream(s1):
            a.add_(1)
        cur.wait_stream(s1)
        cur.wait_stream(s2)
    elif case == "prefork3":      # as prefork, but the nested stream forks/joins several times
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        s3.wait_stream(cur)
        for _ in range(3):
            with torch.cuda.stream(s1):
                a.add_(1)
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                b = a * 2
            with torch.cuda.stream(s1):
                c = a + 5
            s1.wait_stream(s2)
        with torch.cuda.stream(s3):
            d = a * 3
        for s in (s1, s2, s3):
            cur.wait_stream(s)
g.replay()
torch.cuda.synchronize()
print(case, "ok", float(a[0]))
