"""Run the chi=4096 U(1) theta grouped-GEMM plan a few times (for rocprofv3 --pmc passes)."""
import sys
sys.argv = [sys.argv[0], 'none']
exec(open('scripts/first_light.py').read().split("if __name__ == '__main__':")[0])
torch.manual_seed(0)
gemm_perf(4096)
print('probe done')
