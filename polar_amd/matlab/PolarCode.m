classdef PolarCode < handle
    % PolarCode — MATLAB host class over the MI355X polar decoder (C-ABI include/polar_amd.h via the
    % MEX gateway polar_mex.cpp). Same constructor and method names/argument order as the reference
    % class PolarM/PolarCode.m (ctor :59, encode :266, decode_sc_p1 :290, decode_scl_p1 :299,
    % decode_scl_llr :312, get_bler_quick :781), so existing BLER scripts (PolarM/main.m) run unchanged.
    % Row vectors of doubles in and out, like the reference.
    %
    % Build the gateway once:  mex -I<repo>/include polar_mex.cpp -L<repo>/polar_amd -lpolar_amd
    properties
        block_length
        info_length
        crc_size
        n
        design_epsilon
        frozen_bits      % 1 x N, in decoding order (as PolarM)
        info_bits        % 1-based positions of the K+crc unfrozen bits, most reliable first
        crc_matrix
        cc_method        % 'bhattacharya' | 'monte-carlo' (as PolarM :89, :107)
        cc_parameter
        cc_misc
    end
    properties (Access = private)
        h                % uint64 handle owned by the MEX gateway
    end
    methods
        function obj = PolarCode(block_length, info_length, design_epsilon, crc_size)
            if nargin < 4
                crc_size = 0;
            end
            obj.block_length = block_length;
            obj.info_length = info_length;
            obj.n = log2(block_length);
            obj.design_epsilon = design_epsilon;
            obj.crc_size = crc_size;
            obj.h = polar_mex('create', obj.n, info_length, design_epsilon, crc_size);
            [fz, order, crcm] = polar_mex('tables', obj.h);
            obj.frozen_bits = double(fz(:)');
            obj.info_bits = double(order(1:info_length + crc_size)) + 1;
            obj.crc_matrix = double(crcm);
            obj.cc_method = 'bhattacharya';
            obj.cc_parameter = design_epsilon;
            obj.cc_misc = '';
        end
        function monte_carlo_code_construction(obj, design_snr_db, varargin)
            % monte_carlo_code_construction(design_snr_db [, num_runs, constellation_name, receiver_algo, seed])
            % Same method name, argument order and defaults as the reference class (PolarM/PolarCode.m:95); the
            % design itself — genie-aided SC error counting on the GPU, the reference's table cache file, the
            % stable reliability sort and the new decoder handle — is ONE gateway call ('monte_carlo_design').
            opt = [varargin, cell(1, 4 - numel(varargin))];
            defaults = {100e3, 'bpsk', 'bicm', 1};
            unset = cellfun(@isempty, opt);
            opt(unset) = defaults(unset);
            [num_runs, constellation_name, receiver_algo, seed] = opt{:};
            assert(strcmp(receiver_algo, 'bicm'), 'only the bicm receiver is built');
            cid = find(strcmp(constellation_name, {'ask4-gray', 'ask8-gray', 'ask16-gray', 'bpsk'}));   % POLAR_CONST_*
            [obj.cc_method, obj.cc_parameter] = deal('monte-carlo', design_snr_db);
            obj.cc_misc = sprintf('%s_%s_%s', constellation_name, receiver_algo, num2str(num_runs));
            table_file = fullfile('CodeConstructionData', ['MC_block_length_', obj.get_unique_string(), '.txt']);
            old = obj.h;
            [obj.h, fz, order0, est] = polar_mex('monte_carlo_design', obj.n, obj.info_length, obj.crc_size, ...
                uint8(obj.crc_matrix), cid, design_snr_db, num_runs, seed, table_file);
            polar_mex('destroy', old);
            obj.frozen_bits = double(fz);
            obj.info_bits = double(order0(1 : obj.info_length + obj.crc_size)) + 1;
            fprintf('Monte carlo code construction done. Bler estimate = %g\n', est);
        end
        function unique_string = get_unique_string(obj)                % PolarM :258-261
            unique_string = [num2str(obj.block_length), '_', num2str(length(obj.info_bits)), ...
                '_cc_method_', obj.cc_method, '_cc_param_', num2str(obj.cc_parameter), '_', obj.cc_misc];
        end
        function delete(obj)
            if ~isempty(obj.h)
                polar_mex('destroy', obj.h);
                obj.h = [];
            end
        end
        function coded_bits = encode(obj, info_bits)
            coded_bits = double(polar_mex('encode', obj.h, uint8(info_bits(:)')));
        end
        function decoded_bits = decode_sc_p1(obj, p1)
            decoded_bits = double(polar_mex('decode_sc_p1', obj.h, double(p1(:)')));
        end
        function u = decode_scl_p1(obj, p1, p0, list_size)
            u = double(polar_mex('decode_scl_p1', obj.h, double(p1(:)'), double(p0(:)'), list_size));
        end
        function u = decode_scl_llr(obj, llr, list_size, layout)
            % llr may be 1 x N (u is 1 x K, as the reference), B x N (one codeword per row -> u is B x K) or N x B (one codeword
            % per COLUMN -> u is K x B: MATLAB's column-major storage is then the library's and nothing is transposed on either
            % side — the layout for large batches); double or single (single halves the bytes that cross the PCIe link).
            % layout (optional): 'rows' or 'cols' names the layout; it is REQUIRED for a square N x N batch, which fits both
            % (the gateway refuses to guess). One GPU batch; from 32 MiB of LLRs on the library pipelines copy and decode.
            if ~isa(llr, 'single'), llr = double(llr); end
            if nargin < 4
                u = double(polar_mex('decode_scl_llr', obj.h, llr, list_size));
            else
                u = double(polar_mex('decode_scl_llr', obj.h, llr, list_size, layout));
            end
        end
        function [bler, ber] = get_bler_quick(obj, ebno_vec, list_size_vec, max_runs, max_err, seed, devices, constellation_id)
            % [bler, ber] indexed (i_ebno, i_list) as PolarM (:781-850); PolarM constants max_err=50, max_runs=500
            % (:788-789); ber = bit errors per run as the reference computes it (:848). devices (optional): GPU
            % ids to shard the trials over (one process, RCCL all-reduce of the counters). constellation_id (optional):
            % 0 = BPSK, ebno_vec in dB; 1/2/3 = 4/8/16-ASK Gray with BICM, ebno_vec read as the SNR axis in dB
            % (the sweep of main_MC_CC_Comparison.m:44-119).
            if nargin < 4, max_runs = 500; end
            if nargin < 5, max_err = 50; end
            if nargin < 6, seed = 1; end
            if nargin < 7, devices = []; end
            if nargin < 8, constellation_id = 0; end
            [b, e] = polar_mex('get_bler_quick', obj.h, double(ebno_vec(:)'), uint8(list_size_vec(:)'), max_runs, max_err, seed, int32(devices), constellation_id);
            bler = b';      % gateway returns [n_L x n_e] (PolarC layout); PolarM indexes (ebno, list)
            ber = e';
        end
        % names used by the project brief
        function u = decode_SC_P1(obj, p1), u = obj.decode_sc_p1(p1); end
        function u = decode_SCL_P1(obj, p1, p0, L), u = obj.decode_scl_p1(p1, p0, L); end
        function u = decode_SCL_LLR(obj, llr, L), u = obj.decode_scl_llr(llr, L); end
    end
end
