// oracle/driver_main.cpp — TEST INFRASTRUCTURE: the (empty) program of the routed reference drivers.
//
// The reference's matrix-powers harnesses define their kernels in the same file as main
// (mpk/SpM2V.cpp:5-801 + main :804-987; mpk/SpMVmulti0.cpp:22-315 + main :317-418).  To run such a
// driver AS IT IS against the GPU library, oracle/Makefile compiles the reference file where it lies
// into a position-independent shared object — main included, untouched — and links THIS empty
// translation unit against libmpk_mi355.so FIRST and that object second.  The C runtime's start code
// finds `main` in the driver object, and every call the driver makes to a global function (SpM2V_CSR,
// Generate1stlayer, COO2CSR, SpM4V, ...) goes through the PLT and binds to the first definition in
// link order — the shim's — by ordinary ELF symbol interposition: no reference source is edited,
// copied or preprocessed, and the driver's local definitions simply lose.  (g++ keeps such calls
// interposable under -fPIC: it neither inlines nor clones them.)
// tests/test_shim_gpu.py checks with LD_DEBUG=bindings that the kernels did bind to the shim.
