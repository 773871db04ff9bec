// packetkernel_n128.hip -- the packet kernel's 128-byte-node instantiations, a translation unit of their own so that they keep the compiler's default
// instruction scheduling while packetkernel.hip's 64-byte-node ones are built with "max-ilp" (Makefile; packetkernel.hip says what was measured).
#define PT_PK_N128 1
#include "packetkernel.hip"
