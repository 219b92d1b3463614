/* Drop-in for the reference's monolithic/include/parallel_to_grayscale.h:4 -- same prototype, C
 * linkage, implemented by libgrayhip.so on MI355X (see include/grayhip.h for semantics). */
#ifndef PARALLEL_TO_GRAYSCALE_H
#define PARALLEL_TO_GRAYSCALE_H
#ifdef __cplusplus
extern "C" {
#endif
void convert_to_grayscale(unsigned char *data, int width, int height, int channels);
#ifdef __cplusplus
}
#endif
#endif
