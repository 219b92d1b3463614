/* Drop-in for the reference's monolithic/include/sobel.h:3-5 -- same prototype, C linkage,
 * implemented by libgrayhip.so on MI355X.  As in the reference only the interior of dst is
 * written; its border is left untouched. */
#ifndef SOBEL_H
#define SOBEL_H
#ifdef __cplusplus
extern "C" {
#endif
void sobel_edge(const unsigned char *src, unsigned char *dst, int width, int height);
#ifdef __cplusplus
}
#endif
#endif
