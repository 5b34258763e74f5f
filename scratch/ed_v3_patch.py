p='dither_pie_amd/csrc/ediff.hip'; s=open(p).read()
i=s.find("template <int CAP>\n__global__ __launch_bounds__(64 * kMaxWaves) void ed_wavefront_kernel(")
j=s.find("// lane = frame; err rows:")
new=open('scratch/ed_v3_kernel.txt').read()
s=s[:i]+new+s[j:]
s=s.replace("constexpr int kVRing = 16;  // ring of the two boundary rows prefetched from the previous band\nconstexpr int kPixRing = 8;\n","")
s=s.replace("    if (!serpentine && skew * 2 + 2 <= kRing && w + 64 * skew < 65000) {","    if (!serpentine && skew * 2 + 2 <= kRing && w < 60000) {")
s=s.replace("""//   that the producing wave publishes after its stores are acknowledged.  Nothing on the dependency
//   chain touches HBM latency: errors live in an 8-deep per-row LDS ring, the two rows that cross a
//   band boundary travel through a global row buffer (L2) and are prefetched one step ahead into a
//   small LDS ring, and input pixels are prefetched two steps ahead into an LDS ring.""","""//   that the producing wave publishes after its stores are acknowledged.  Nothing on the dependency
//   chain touches global memory: all global traffic happens at 16-step period boundaries -- each lane
//   fetches the 48 bytes of its next 16 pixels one full period ahead (unaligned dword loads), flushes
//   the 48 output bytes of the finished period, rows 62/63 flush their staged errors to a global row
//   buffer (L2) and lanes prefetch the previous band's two boundary rows into a 64-column LDS ring --
//   so every wait is for operations issued a period earlier.  Errors of the band's own rows live in an
//   8-deep per-row LDS ring.""")
open(p,'w').write(s)
print('patched')
