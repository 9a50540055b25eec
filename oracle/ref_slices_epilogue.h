// C wrappers around the reference slices (see oracle/Makefile for the line ranges).
extern "C" {
int ref_getOptimalBufferSize(int fps) { return livim::getOptimalBufferSize(fps); }
int ref_butterworth(unsigned N, double Wn, double* a, double* b) {
    std::vector<double> va, vb;
    livim::butterworth(N, Wn, va, vb);
    for (size_t i = 0; i < va.size(); ++i) a[i] = va[i];
    for (size_t i = 0; i < vb.size(); ++i) b[i] = vb[i];
    return (int)va.size();
}
double ref_motionHzToBlend(double hz, double fps) { return livim::motionHzToBlend(hz, fps); }
}
