// integration/empty_main.cpp — the (empty) program of a routed reference driver (integration/mpk_mi355.mk).
//
// mpk/SpM2V.cpp and mpk/SpMVmulti0.cpp define their kernels in the same file as main (SpM2V.cpp:5-801 + main :804-987;
// SpMVmulti0.cpp:22-315 + main :317-418).  To run such a driver AS IT IS against the GPU library, the reference file is
// compiled into a position-independent shared object — main included, untouched — and THIS empty translation unit is linked
// against libmpk_mi355.so FIRST and that object second: the C runtime's start code finds `main` in the driver object, and
// every call the driver makes to a global function (SpM2V_CSR, Generate1stlayer, COO2CSR, SpM4V, ...) goes through the PLT and
// binds to the first definition in link order — the shim's — by ordinary ELF symbol interposition.
