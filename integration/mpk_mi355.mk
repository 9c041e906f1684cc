# integration/mpk_mi355.mk — include this from the reference's mpk/Makefile (one line: `include /path/to/repo/integration/mpk_mi355.mk`)
# to get GPU twins of its targets (mpk/Makefile:16-23: spmv, 2spmv, spm2v) WITHOUT editing a source file:
#
#     make 2spmv_mi355 spm2v_mi355 multi0_mi355        # the same drivers, their kernels served by libmpk_mi355.so
#     ./2spmv_mi355 mat/matrix10_aij.mtx               # prints the same "name : us | xspeed-up | rel err" lines (mpk/2SpMV.cpp:146-293)
#     sh $(MI355_REPO)/integration/run_mpk_mi355.sh    # the loop of mpk/SpMV.sh / SpM2V.sh over mat/matrix{1..10}_aij.mtx
#
# 2SpMV.cpp calls only functions declared in mpk/SpMV.h: it links against the shim in the place of SpMV.cpp + utils.cpp.
# SpM2V.cpp and SpMVmulti0.cpp define their kernels beside main: they are built as position-independent shared objects as
# they are, and an empty program links the shim FIRST — the driver's calls then bind to the shim (ELF symbol interposition;
# INTEGRATION.md §2, integration/empty_main.cpp).  CXX / CXXFLAGS are the reference's own (mpk/Makefile:9-10).
MI355_REPO ?= $(abspath $(dir $(lastword $(MAKEFILE_LIST)))/..)
MI355_LIB  = $(MI355_REPO)/navierstokes_amd/csrc
MI355_LINK = -L$(MI355_LIB) -lmpk_mi355 -lmi355spmv -Wl,-rpath,$(MI355_LIB)

2spmv_mi355: 2SpMV.cpp SpMV.h
	$(CXX) $(CXXFLAGS) -o $@ 2SpMV.cpp $(MI355_LINK)

libdrv_spm2v.so: SpM2V.cpp SpMV.h
	$(CXX) $(CXXFLAGS) -U_FORTIFY_SOURCE -fPIC -shared -o $@ SpM2V.cpp
spm2v_mi355: libdrv_spm2v.so
	$(CXX) -o $@ $(MI355_REPO)/integration/empty_main.cpp -Wl,--no-as-needed $(MI355_LINK) -L. -ldrv_spm2v -Wl,-rpath,'$$ORIGIN'

libdrv_multi0.so: SpMVmulti0.cpp
	$(CXX) $(CXXFLAGS) -U_FORTIFY_SOURCE -fPIC -shared -o $@ SpMVmulti0.cpp
multi0_mi355: libdrv_multi0.so
	$(CXX) -o $@ $(MI355_REPO)/integration/empty_main.cpp -Wl,--no-as-needed $(MI355_LINK) -L. -ldrv_multi0 -Wl,-rpath,'$$ORIGIN'

clean_mi355:
	rm -f 2spmv_mi355 spm2v_mi355 multi0_mi355 libdrv_spm2v.so libdrv_multi0.so
