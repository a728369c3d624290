# Builds everything __graft_entry__.build() builds, for a maintainer without Python in the loop (INTEGRATION.md tells
# upstream's CMake to find_library(iyokan_hip) in iyokan_amd/lib):
#   make            libiyokan_hip.so (the product), libiyokan_client.so, libiyk_emul.so, the oracle, the C++ host self-test
#   make lib        only iyokan_amd/lib/libiyokan_hip.so
HIPCC ?= hipcc
CXX ?= g++
ARCH ?= gfx950
CSRC := iyokan_amd/csrc
LIBDIR := iyokan_amd/lib
HDRS := $(wildcard $(CSRC)/*.hpp) $(wildcard include/*.h)
HOST_FLAGS := -O3 -march=x86-64-v3 -ffp-contract=off -std=c++17 -fPIC -shared

.PHONY: all lib host oracle clean
all: lib $(LIBDIR)/libiyokan_client.so $(LIBDIR)/libiyk_emul.so oracle host
lib: $(LIBDIR)/libiyokan_hip.so

$(LIBDIR)/libiyokan_hip.so: $(CSRC)/iyokan_hip.hip $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DIYK_BUILD_ID='"$(shell python3 tools/src_hash.py)"' -o $@ $<
$(LIBDIR)/libiyokan_client.so: $(CSRC)/client.cpp $(HDRS)
	@mkdir -p $(LIBDIR)
	$(CXX) $(HOST_FLAGS) -o $@ $<
$(LIBDIR)/libiyk_emul.so: $(CSRC)/emul.cpp $(HDRS)
	@mkdir -p $(LIBDIR)
	$(CXX) $(HOST_FLAGS) -o $@ $<
oracle:
	$(MAKE) -C oracle
host: lib $(LIBDIR)/libiyokan_client.so
	$(MAKE) -C iyokan_amd/host
clean:
	rm -f $(LIBDIR)/*.so iyokan_amd/host/test0_hip oracle/libiyk_oracle.so
